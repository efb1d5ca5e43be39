/*
 * nnbvh_oracle.c — TEST INFRASTRUCTURE ONLY (the parity oracle); see nnbvh_oracle.h.
 *
 * Scalar C restatement of the reference hot path, arithmetic operation by
 * arithmetic operation:
 *   traversal        /root/reference/src/pbrt/cpu/aggregates.cpp:529-579 (closest), 581-624 (any)
 *   slab test        /root/reference/src/pbrt/util/vecmath.h:1573-1608
 *   triangle test    /root/reference/src/pbrt/shapes.cpp:172-273 (called from 320-358)
 *   patch test       /root/reference/src/pbrt/shapes.h:1279-1347, util/math.h:614-637, 1420-1426
 *   helpers          util/math.h:569-575 (DifferenceOfProducts), util/float.h:100-102 (FMA),
 *                    util/float.h:195-197 (gamma), util/vecmath.h:999-1004 (Cross), 964-967 (Dot)
 *
 * Build contract: -ffp-contract=off (the reference disables contraction,
 * CMakeLists.txt:134-137); fmaf() appears exactly where the reference calls FMA().
 *
 * Pinning: the leaf tests and the slab test are checked bit-for-bit against the
 * reference's own compiled functions (oracle/_ref/ref_leaf, built by oracle/Makefile
 * from the sources under /root/reference) and against tests/golden/ vectors
 * generated from that binary.  The traversal LOOP itself cannot be compiled from the
 * reference here (Primitive::Intersect lives in cpu/primitive.cpp, which needs the
 * un-vendored nanovdb header), so its node-visit counts are "parity unpinned" beyond
 * this restatement, the brute-force cross-check and the aggregate figures recorded in
 * SURVEY.md §6 — see DESIGN.md §Oracle.
 */
#include "nnbvh_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- util/float.h:43,195-197 -------------------------------------------------------- */
#define ORC_EPS 0x1p-24f /* MachineEpsilon = numeric_limits<float>::epsilon() * 0.5 */
static inline float orc_gamma(int n) {
    float ne = (float)n * ORC_EPS;
    return ne / (1.0f - ne);
}

/* ---- util/math.h:569-575 ------------------------------------------------------------ */
static inline float orc_dop(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = fmaf(a, b, -cd);
    float err = fmaf(-c, d, cd);
    return diff + err;
}

static inline float orc_max3(float a, float b, float c) {
    /* std::max({a,b,c}): left fold with operator< */
    float m = a;
    if (m < b) m = b;
    if (m < c) m = c;
    return m;
}

/* util/vecmath.h:999-1004 */
static inline void orc_cross(const float v[3], const float w[3], float out[3]) {
    out[0] = orc_dop(v[1], w[2], v[2], w[1]);
    out[1] = orc_dop(v[2], w[0], v[0], w[2]);
    out[2] = orc_dop(v[0], w[1], v[1], w[0]);
}
static inline float orc_dot(const float v[3], const float w[3]) {
    return v[0] * w[0] + v[1] * w[1] + v[2] * w[2];
}
static inline float orc_len2(const float v[3]) {
    return v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
}

/* ---- slab test: util/vecmath.h:1573-1608 -------------------------------------------- */
static inline int slab_test(const float *pmin, const float *pmax, const float o[3],
                            float ray_tmax, const float inv[3], const int neg[3]) {
    const float *b[2] = {pmin, pmax};
    const float widen = 1.0f + 2.0f * orc_gamma(3);
    float tmin = (b[neg[0]][0] - o[0]) * inv[0];
    float tmax = (b[1 - neg[0]][0] - o[0]) * inv[0];
    float tymin = (b[neg[1]][1] - o[1]) * inv[1];
    float tymax = (b[1 - neg[1]][1] - o[1]) * inv[1];
    tmax *= widen;
    tymax *= widen;
    if (tmin > tymax || tymin > tmax) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b[neg[2]][2] - o[2]) * inv[2];
    float tzmax = (b[1 - neg[2]][2] - o[2]) * inv[2];
    tzmax *= widen;
    if (tmin > tzmax || tzmin > tmax) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < ray_tmax) && (tmax > 0.0f);
}

int orc_slab(const float bounds[6], const float o[3], const float d[3], float ray_tmax) {
    /* invDir / dirIsNeg as the traversal prepares them, aggregates.cpp:534-535 */
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    int neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
    return slab_test(bounds, bounds + 3, o, ray_tmax, inv, neg);
}

/* ---- triangle: shapes.cpp:172-273 --------------------------------------------------- */
int orc_triangle(const float o[3], const float d[3], float tmax, const float p0[3],
                 const float p1[3], const float p2[3], float out[4]) {
    /* degenerate-triangle rejection, :176-177 */
    float e02[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    float e01[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    float cr[3];
    orc_cross(e02, e01, cr);
    if (orc_len2(cr) == 0.0f) return 0;

    /* translate to ray origin, :181-183 */
    float a[3] = {p0[0] - o[0], p0[1] - o[1], p0[2] - o[2]};
    float b[3] = {p1[0] - o[0], p1[1] - o[1], p1[2] - o[2]};
    float c[3] = {p2[0] - o[0], p2[1] - o[1], p2[2] - o[2]};

    /* permutation, :186-196; MaxComponentIndex vecmath.h:453-455 */
    float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    int kx = kz + 1;
    if (kx == 3) kx = 0;
    int ky = kx + 1;
    if (ky == 3) ky = 0;
    float dx = d[kx], dy = d[ky], dz = d[kz];
    float p0x = a[kx], p0y = a[ky], p0z = a[kz];
    float p1x = b[kx], p1y = b[ky], p1z = b[kz];
    float p2x = c[kx], p2y = c[ky], p2z = c[kz];

    /* shear, :199-207 (plain mul + add, no FMA) */
    float sx = -dx / dz, sy = -dy / dz, sz = 1.0f / dz;
    p0x += sx * p0z;
    p0y += sy * p0z;
    p1x += sx * p1z;
    p1y += sy * p1z;
    p2x += sx * p2z;
    p2y += sy * p2z;

    /* edge functions, :210-212 */
    float e0 = orc_dop(p1x, p2y, p1y, p2x);
    float e1 = orc_dop(p2x, p0y, p2y, p0x);
    float e2 = orc_dop(p0x, p1y, p0y, p1x);

    /* double-precision fallback on an exact zero, :215-225 */
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2x * (double)p1y;
        double p2typ1tx = (double)p2y * (double)p1x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0x * (double)p2y;
        double p0typ2tx = (double)p0y * (double)p2x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1x * (double)p0y;
        double p1typ0tx = (double)p1y * (double)p0x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }

    /* sign and determinant tests, :228-232 */
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return 0;
    float det = e0 + e1 + e2;
    if (det == 0) return 0;

    /* scaled distance vs range, :235-242 */
    p0z *= sz;
    p1z *= sz;
    p2z *= sz;
    float tscaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0 && (tscaled >= 0 || tscaled < tmax * det)) return 0;
    else if (det > 0 && (tscaled <= 0 || tscaled > tmax * det)) return 0;

    /* barycentrics and t, :245-247 */
    float invdet = 1.0f / det;
    float b0 = e0 * invdet, b1 = e1 * invdet, b2 = e2 * invdet;
    float t = tscaled * invdet;

    /* conservative t > 0 bound, :251-269 */
    float maxzt = orc_max3(fabsf(p0z), fabsf(p1z), fabsf(p2z));
    float deltaz = orc_gamma(3) * maxzt;
    float maxxt = orc_max3(fabsf(p0x), fabsf(p1x), fabsf(p2x));
    float maxyt = orc_max3(fabsf(p0y), fabsf(p1y), fabsf(p2y));
    float deltax = orc_gamma(5) * (maxxt + maxzt);
    float deltay = orc_gamma(5) * (maxyt + maxzt);
    float deltae = 2.0f * (orc_gamma(2) * maxxt * maxyt + deltay * maxxt + deltax * maxyt);
    float maxe = orc_max3(fabsf(e0), fabsf(e1), fabsf(e2));
    float deltat =
        3.0f * (orc_gamma(3) * maxe * maxzt + deltae * maxzt + deltaz * maxe) * fabsf(invdet);
    if (t <= deltat) return 0;

    out[0] = b0;
    out[1] = b1;
    out[2] = b2;
    out[3] = t;
    return 1;
}

/* ---- Quadratic: util/math.h:614-637 ------------------------------------------------- */
static int orc_quadratic(float a, float b, float c, float *t0, float *t1) {
    if (a == 0) {
        if (b == 0) return 0;
        *t0 = *t1 = -c / b;
        return 1;
    }
    float discrim = orc_dop(b, b, 4.0f * a, c);
    if (discrim < 0) return 0;
    float root = sqrtf(discrim);
    float q = -0.5f * (b + copysignf(root, b));
    *t0 = q / a;
    *t1 = c / q;
    if (*t0 > *t1) {
        float s = *t0;
        *t0 = *t1;
        *t1 = s;
    }
    return 1;
}

/* Determinant(SquareMatrix<3>): util/math.h:1420-1426; m row-major */
static inline float orc_det3(const float m[9]) {
    float minor12 = orc_dop(m[4], m[8], m[5], m[7]);
    float minor02 = orc_dop(m[3], m[8], m[5], m[6]);
    float minor01 = orc_dop(m[3], m[7], m[4], m[6]);
    return fmaf(m[2], minor01, orc_dop(m[0], minor12, m[1], minor02));
}

static inline float orc_maxabs3(const float v[3]) {
    return orc_max3(fabsf(v[0]), fabsf(v[1]), fabsf(v[2]));
}

/* Lerp(t, a, b) = (1 - t) * a + t * b, util/vecmath.h:410-412 */
static inline void orc_lerp3(float t, const float a[3], const float b[3], float out[3]) {
    float omt = 1.0f - t;
    for (int k = 0; k < 3; ++k) out[k] = omt * a[k] + t * b[k];
}

/* one root's (v numerator, t numerator, p2): shapes.h:1303-1316 */
static inline void blp_root(float u, const float o[3], const float d[3], const float p00[3],
                            const float p10[3], const float p01[3], const float p11[3],
                            float *vnum, float *tnum, float *p2) {
    float uo[3], l1[3], ud[3], deltao[3], perp[3];
    orc_lerp3(u, p00, p10, uo);
    orc_lerp3(u, p01, p11, l1);
    for (int k = 0; k < 3; ++k) ud[k] = l1[k] - uo[k];
    for (int k = 0; k < 3; ++k) deltao[k] = uo[k] - o[k];
    orc_cross(d, ud, perp);
    *p2 = orc_len2(perp);
    float mv[9] = {deltao[0], d[0], perp[0], deltao[1], d[1], perp[1], deltao[2], d[2], perp[2]};
    float mt[9] = {deltao[0], ud[0], perp[0], deltao[1], ud[1], perp[1],
                   deltao[2], ud[2], perp[2]};
    *vnum = orc_det3(mv);
    *tnum = orc_det3(mt);
}

/* ---- bilinear patch: shapes.h:1279-1347 --------------------------------------------- */
int orc_bilinear_patch(const float o[3], const float d[3], float tmax, const float p00[3],
                       const float p10[3], const float p01[3], const float p11[3],
                       float out[3]) {
    float e10[3], e0111[3], e0100[3], e1110[3], po0[3], po1[3], cr[3];
    for (int k = 0; k < 3; ++k) {
        e10[k] = p10[k] - p00[k];
        e0111[k] = p01[k] - p11[k];
        e0100[k] = p01[k] - p00[k];
        e1110[k] = p11[k] - p10[k];
        po0[k] = p00[k] - o[k];
        po1[k] = p10[k] - o[k];
    }
    orc_cross(e10, e0111, cr);
    float a = orc_dot(cr, d);
    orc_cross(po0, d, cr);
    float c = orc_dot(cr, e0100);
    orc_cross(po1, d, cr);
    float b = orc_dot(cr, e1110) - (a + c);

    float u1, u2;
    if (!orc_quadratic(a, b, c, &u1, &u2)) return 0;

    float eps = orc_gamma(10) * (orc_maxabs3(o) + orc_maxabs3(d) + orc_maxabs3(p00) +
                                 orc_maxabs3(p10) + orc_maxabs3(p01) + orc_maxabs3(p11));

    float t = tmax, u = 0, v = 0;
    if (0 <= u1 && u1 <= 1) {
        float v1, t1, p2;
        blp_root(u1, o, d, p00, p10, p01, p11, &v1, &t1, &p2);
        if (t1 > p2 * eps && 0 <= v1 && v1 <= p2) {
            u = u1;
            v = v1 / p2;
            t = t1 / p2;
        }
    }
    if (0 <= u2 && u2 <= 1 && u2 != u1) {
        float v2, t2, p2;
        blp_root(u2, o, d, p00, p10, p01, p11, &v2, &t2, &p2);
        t2 /= p2;
        if (0 <= v2 && v2 <= p2 && t > t2 && t2 > eps) {
            t = t2;
            u = u2;
            v = v2 / p2;
        }
    }
    if (t >= tmax) return 0;
    out[0] = u;
    out[1] = v;
    out[2] = t;
    return 1;
}

/* ---- batched single-function forms -------------------------------------------------- */
void orc_slab_batch(const float *bounds6, const float *o3, const float *d3, const float *tmax,
                    int n, uint8_t *out) {
    for (int i = 0; i < n; ++i)
        out[i] = (uint8_t)orc_slab(bounds6 + 6 * i, o3 + 3 * i, d3 + 3 * i, tmax[i]);
}
void orc_triangle_batch(const float *o3, const float *d3, const float *tmax, const float *p9,
                        int n, uint8_t *hit, float *out4) {
    for (int i = 0; i < n; ++i) {
        float r[4] = {0, 0, 0, 0};
        hit[i] = (uint8_t)orc_triangle(o3 + 3 * i, d3 + 3 * i, tmax[i], p9 + 9 * i,
                                       p9 + 9 * i + 3, p9 + 9 * i + 6, r);
        memcpy(out4 + 4 * i, r, sizeof r);
    }
}
void orc_bilinear_patch_batch(const float *o3, const float *d3, const float *tmax,
                              const float *p12, int n, uint8_t *hit, float *out3) {
    for (int i = 0; i < n; ++i) {
        float r[3] = {0, 0, 0};
        hit[i] = (uint8_t)orc_bilinear_patch(o3 + 3 * i, d3 + 3 * i, tmax[i], p12 + 12 * i,
                                             p12 + 12 * i + 3, p12 + 12 * i + 6,
                                             p12 + 12 * i + 9, r);
        memcpy(out3 + 3 * i, r, sizeof r);
    }
}

/* ---- the ray transform of TransformedPrimitive -------------------------------------------
 * Transform::ApplyInverse(const Ray&, Float *tMax)   util/transform.h:416-429
 * Transform::ApplyInverse(const Point3fi&)           util/transform.cpp:263-303 (exact-input branch)
 * Interval::FromValueAndError, operator+=, Midpoint, Width   util/math.h:829-853, 874-876, 905-908
 * NextFloatUp/Down, Add/Sub/DivRound* (CPU forms)    util/float.h:163-193, 199-260 */
typedef struct {
    float lo, hi;
} orc_ivl;

static inline float next_up(float v) {
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui;
    memcpy(&ui, &v, 4);
    if (v >= 0) ++ui;
    else --ui;
    memcpy(&v, &ui, 4);
    return v;
}
static inline float next_down(float v) {
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui;
    memcpy(&ui, &v, 4);
    if (v > 0) --ui;
    else ++ui;
    memcpy(&v, &ui, 4);
    return v;
}
static inline orc_ivl ivl(float a, float b) { /* Interval(low, high): min / max, math.h:825-826 */
    orc_ivl r;
    r.lo = (b < a) ? b : a;
    r.hi = (a < b) ? b : a;
    return r;
}
static inline orc_ivl ivl_from_value_and_error(float v, float err) { /* math.h:829-838 */
    orc_ivl i;
    if (err == 0) i.lo = i.hi = v;
    else {
        i.lo = next_down(v + (-err)); /* SubRoundDown(v, err) = AddRoundDown(v, -err) */
        i.hi = next_up(v + err);
    }
    return i;
}
static inline orc_ivl ivl_add_f(orc_ivl a, float f) { /* `x += v.x`: math.h:922 -> 905-908 -> 874-876 */
    return ivl(next_down(a.lo + f), next_up(a.hi + f));
}
static inline orc_ivl ivl_div_f(orc_ivl i, float f) { /* math.h:1028-1037 */
    if (f == 0) return ivl(-INFINITY, INFINITY);
    if (f > 0) return ivl(next_down(i.lo / f), next_up(i.hi / f));
    return ivl(next_down(i.hi / f), next_up(i.lo / f));
}

void orc_apply_inverse_ray(const float mi[12], const float o[3], const float d[3], float tmax,
                           float out[7]) {
    const float x = o[0], y = o[1], z = o[2]; /* Float(p.x): midpoint of an exact interval */
    const float g3 = orc_gamma(3);
    float p[3], e[3];
    for (int k = 0; k < 3; ++k) {
        const float *r = mi + 4 * k;
        p[k] = (r[0] * x + r[1] * y) + (r[2] * z + r[3]);
        e[k] = g3 * (fabsf(r[0] * x) + fabsf(r[1] * y) + fabsf(r[2] * z));
    }
    const float wp = (0.f * x + 0.f * y) + (0.f * z + 1.f); /* row 3 of an affine inverse */
    orc_ivl xp = ivl_from_value_and_error(p[0], e[0]);
    orc_ivl yp = ivl_from_value_and_error(p[1], e[1]);
    orc_ivl zp = ivl_from_value_and_error(p[2], e[2]);
    if (!(wp == 1)) {
        xp = ivl_div_f(xp, wp);
        yp = ivl_div_f(yp, wp);
        zp = ivl_div_f(zp, wp);
    }
    /* Transform::ApplyInverse(Vector3f), transform.h:401-406 */
    float dx = mi[0] * d[0] + mi[1] * d[1] + mi[2] * d[2];
    float dy = mi[4] * d[0] + mi[5] * d[1] + mi[6] * d[2];
    float dz = mi[8] * d[0] + mi[9] * d[1] + mi[10] * d[2];
    float len2 = dx * dx + dy * dy + dz * dz;
    if (len2 > 0) { /* transform.h:420-427 */
        float ex = (xp.hi - xp.lo) / 2, ey = (yp.hi - yp.lo) / 2, ez = (zp.hi - zp.lo) / 2;
        float dt = (fabsf(dx) * ex + fabsf(dy) * ey + fabsf(dz) * ez) / len2;
        xp = ivl_add_f(xp, dx * dt);
        yp = ivl_add_f(yp, dy * dt);
        zp = ivl_add_f(zp, dz * dt);
        tmax -= dt;
    }
    out[0] = (xp.lo + xp.hi) / 2; /* Point3f(o): Interval::Midpoint, math.h:851 */
    out[1] = (yp.lo + yp.hi) / 2;
    out[2] = (zp.lo + zp.hi) / 2;
    out[3] = dx;
    out[4] = dy;
    out[5] = dz;
    out[6] = tmax;
}

void orc_apply_inverse_ray_batch(const float *mi12, const float *o3, const float *d3,
                                 const float *tmax, int n, float *out7) {
    for (int i = 0; i < n; ++i)
        orc_apply_inverse_ray(mi12 + 12 * i, o3 + 3 * i, d3 + 3 * i, tmax[i], out7 + 7 * i);
}

/* Transform::operator()(const Bounds3f&), util/transform.cpp:134-139, with the point
 * transform of util/transform.h:310-319 and Bounds3::Corner (vecmath.h:1284-1289) */
void orc_transform_bounds(const float m[12], const float in[6], float out[6]) {
    float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    float mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    for (int c = 0; c < 8; ++c) {
        float p[3] = {in[(c & 1) ? 3 : 0], in[(c & 2) ? 4 : 1], in[(c & 4) ? 5 : 2]};
        for (int k = 0; k < 3; ++k) {
            float v = m[4 * k] * p[0] + m[4 * k + 1] * p[1] + m[4 * k + 2] * p[2] + m[4 * k + 3];
            if (v < mn[k]) mn[k] = v;
            if (mx[k] < v) mx[k] = v;
        }
    }
    memcpy(out, mn, 12);
    memcpy(out + 3, mx, 12);
}

/* ---- AnimatedTransform::Interpolate, util/transform.cpp:1062-1081 ------------------------------ */
static void mat4_mul(const float a[16], const float b[16], float r[16]) { /* math.h:1499-1509 */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0;
            for (int k = 0; k < 4; ++k) acc = fmaf(a[4 * i + k], b[4 * k + j], acc);
            r[4 * i + j] = acc;
        }
}

/* internal::InnerProduct / TwoProd / TwoSum, util/math.h:585-612 and float.h compensated arithmetic */
typedef struct {
    float v, err;
} orc_cf;
static orc_cf two_prod(float a, float b) {
    float ab = a * b;
    orc_cf r = {ab, fmaf(a, b, -ab)};
    return r;
}
static orc_cf two_sum(float a, float b) {
    float s = a + b, delta = s - a;
    orc_cf r = {s, (a - (s - delta)) + (b - delta)};
    return r;
}
static orc_cf inner_rec(const float *t, int n) { /* n pairs */
    orc_cf ab = two_prod(t[0], t[1]);
    if (n == 1) return ab;
    orc_cf tp = inner_rec(t + 2, n - 1);
    orc_cf sum = two_sum(ab.v, tp.v);
    orc_cf r = {sum.v, ab.err + (tp.err + sum.err)};
    return r;
}
static float inner3(float a0, float b0, float a1, float b1, float a2, float b2) {
    const float t[6] = {a0, b0, a1, b1, a2, b2};
    orc_cf ip = inner_rec(t, 3);
    return ip.v + ip.err;
}
static float inner6(const float t[12]) {
    orc_cf ip = inner_rec(t, 6);
    return ip.v + ip.err;
}

/* Inverse(const SquareMatrix<4> &), util/math.h:1572-1625; returns 0 for a singular matrix */
static int mat4_inverse(const float mm[16], float out[16]) {
#define M(i, j) mm[4 * (i) + (j)]
    float s0 = orc_dop(M(0, 0), M(1, 1), M(1, 0), M(0, 1));
    float s1 = orc_dop(M(0, 0), M(1, 2), M(1, 0), M(0, 2));
    float s2 = orc_dop(M(0, 0), M(1, 3), M(1, 0), M(0, 3));
    float s3 = orc_dop(M(0, 1), M(1, 2), M(1, 1), M(0, 2));
    float s4 = orc_dop(M(0, 1), M(1, 3), M(1, 1), M(0, 3));
    float s5 = orc_dop(M(0, 2), M(1, 3), M(1, 2), M(0, 3));
    float c0 = orc_dop(M(2, 0), M(3, 1), M(3, 0), M(2, 1));
    float c1 = orc_dop(M(2, 0), M(3, 2), M(3, 0), M(2, 2));
    float c2 = orc_dop(M(2, 0), M(3, 3), M(3, 0), M(2, 3));
    float c3 = orc_dop(M(2, 1), M(3, 2), M(3, 1), M(2, 2));
    float c4 = orc_dop(M(2, 1), M(3, 3), M(3, 1), M(2, 3));
    float c5 = orc_dop(M(2, 2), M(3, 3), M(3, 2), M(2, 3));
    const float dt[12] = {s0, c5, -s1, c4, s2, c3, s3, c2, s5, c0, -s4, c1};
    float determinant = inner6(dt);
    if (determinant == 0) return 0;
    float s = 1 / determinant;
    out[0] = s * inner3(M(1, 1), c5, M(1, 3), c3, -M(1, 2), c4);
    out[1] = s * inner3(-M(0, 1), c5, M(0, 2), c4, -M(0, 3), c3);
    out[2] = s * inner3(M(3, 1), s5, M(3, 3), s3, -M(3, 2), s4);
    out[3] = s * inner3(-M(2, 1), s5, M(2, 2), s4, -M(2, 3), s3);
    out[4] = s * inner3(-M(1, 0), c5, M(1, 2), c2, -M(1, 3), c1);
    out[5] = s * inner3(M(0, 0), c5, M(0, 3), c1, -M(0, 2), c2);
    out[6] = s * inner3(-M(3, 0), s5, M(3, 2), s2, -M(3, 3), s1);
    out[7] = s * inner3(M(2, 0), s5, M(2, 3), s1, -M(2, 2), s2);
    out[8] = s * inner3(M(1, 0), c4, M(1, 3), c0, -M(1, 1), c2);
    out[9] = s * inner3(-M(0, 0), c4, M(0, 1), c2, -M(0, 3), c0);
    out[10] = s * inner3(M(3, 0), s4, M(3, 3), s0, -M(3, 1), s2);
    out[11] = s * inner3(-M(2, 0), s4, M(2, 1), s2, -M(2, 3), s0);
    out[12] = s * inner3(-M(1, 0), c3, M(1, 1), c1, -M(1, 2), c0);
    out[13] = s * inner3(M(0, 0), c3, M(0, 2), c0, -M(0, 1), c1);
    out[14] = s * inner3(-M(3, 0), s3, M(3, 1), s1, -M(3, 2), s0);
    out[15] = s * inner3(M(2, 0), s3, M(2, 2), s0, -M(2, 1), s1);
#undef M
    return 1;
}

/* 0 = libm's sinf, as the reference; 1 = sine evaluated in double and rounded once, as the device's
 * Slerp does (the documented tolerance exception of AnimatedPrimitive).  Only the per-ray
 * sines of Interpolate are switched; theta and SinXOverX(theta) are host-side libm values in both. */
static int g_sin_mode = 0;
void orc_set_sin_mode(int mode) { g_sin_mode = mode; }
static float sin_x_over_x_ray(float x) { /* util/math.h:340-344 */
    if (1 - x * x == 1) return 1;
    return (g_sin_mode ? (float)sin((double)x) : sinf(x)) / x;
}
static float sin_x_over_x(float x) { /* util/math.h:340-344 */
    if (1 - x * x == 1) return 1;
    return sinf(x) / x;
}
static float quat_dot(const float a[4], const float b[4]) { /* vecmath.h:1126-1128, 964-967 */
    return (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) + a[3] * b[3];
}
static float quat_angle_between(const float q1[4], const float q2[4]) { /* vecmath.h:1138-1143 */
    float t[4];
    if (quat_dot(q1, q2) < 0) {
        for (int k = 0; k < 4; ++k) t[k] = q1[k] + q2[k];
        float x = sqrtf(quat_dot(t, t)) / 2;
        x = x < -1 ? -1 : (x > 1 ? 1 : x); /* SafeASin: Clamp(x, -1, 1) */
        return 3.14159265358979323846f - 2 * asinf(x);
    }
    for (int k = 0; k < 4; ++k) t[k] = q2[k] - q1[k];
    float x = sqrtf(quat_dot(t, t)) / 2;
    x = x < -1 ? -1 : (x > 1 ? 1 : x);
    return 2 * asinf(x);
}

void orc_anim_interpolate(const orc_anim *a, float time, float m[16], float minv[16]) {
    if (!a->actually_animated || time <= a->start_time) { /* :1064-1065 */
        memcpy(m, a->start_m, 64);
        memcpy(minv, a->start_minv, 64);
        return;
    }
    if (time >= a->end_time) { /* :1066-1067 */
        memcpy(m, a->end_m, 64);
        memcpy(minv, a->end_minv, 64);
        return;
    }
    const float dt = (time - a->start_time) / (a->end_time - a->start_time);
    float trans[3];
    for (int k = 0; k < 3; ++k) trans[k] = (1 - dt) * a->T[0][k] + dt * a->T[1][k];
    /* Slerp(dt, R[0], R[1]), vecmath.h:1146-1151 */
    const float theta = quat_angle_between(a->R[0], a->R[1]);
    const float sinThetaOverTheta = sin_x_over_x(theta);
    const float w1 = sin_x_over_x_ray((1 - dt) * theta), w2 = sin_x_over_x_ray(dt * theta);
    float q[4];
    for (int k = 0; k < 4; ++k)
        q[k] = a->R[0][k] * (1 - dt) * w1 / sinThetaOverTheta + a->R[1][k] * dt * w2 / sinThetaOverTheta;
    float scale[16];
    for (int k = 0; k < 16; ++k) scale[k] = a->S[0][k] * (1 - dt) + a->S[1][k] * dt; /* (1-dt)*S0 + dt*S1 = S*s */
    /* Translate(trans), transform.cpp:21-31 */
    float tm[16] = {1, 0, 0, trans[0], 0, 1, 0, trans[1], 0, 0, 1, trans[2], 0, 0, 0, 1};
    float tmi[16] = {1, 0, 0, -trans[0], 0, 1, 0, -trans[1], 0, 0, 1, -trans[2], 0, 0, 0, 1};
    /* Transform(Quaternion q), transform.h:367-385: mInv from q, m = Transpose(mInv) */
    const float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2];
    const float xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const float wx = q[0] * q[3], wy = q[1] * q[3], wz = q[2] * q[3];
    float rmi[16] = {1 - 2 * (yy + zz), 2 * (xy + wz), 2 * (xz - wy), 0,
                     2 * (xy - wz), 1 - 2 * (xx + zz), 2 * (yz + wx), 0,
                     2 * (xz + wy), 2 * (yz - wx), 1 - 2 * (xx + yy), 0,
                     0, 0, 0, 1};
    float rm[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) rm[4 * i + j] = rmi[4 * j + i];
    /* Transform(scale): mInv = Inverse(scale), NaN-filled when singular (transform.h:64-75) */
    float smi[16];
    if (!mat4_inverse(scale, smi))
        for (int k = 0; k < 16; ++k) smi[k] = NAN;
    /* Translate * Rotate * Scale: (A * B).m = A.m * B.m, .mInv = B.mInv * A.mInv (transform.cpp:141-143) */
    float tr[16], tri[16];
    mat4_mul(tm, rm, tr);
    mat4_mul(rmi, tmi, tri);
    mat4_mul(tr, scale, m);
    mat4_mul(smi, tri, minv);
}

void orc_anim_interpolate_batch(const orc_anim *a, const float *time, int n, float *out32) {
    for (int i = 0; i < n; ++i) orc_anim_interpolate(&a[i], time[i], out32 + 32 * i, out32 + 32 * i + 16);
}

/* the animation table and the current ray's time for the two-level traversal below (per worker thread) */
static __thread const orc_anim *g_anims = NULL;
static __thread float g_ray_time = 0.0f;
/* the 3x4 inverse matrix AnimatedPrimitive / TransformedPrimitive apply to the ray */
static void instance_minv(const orc_instance *in, int index, float out12[12]) {
    if (g_anims && g_anims[index].actually_animated) {
        float m[16], mi[16];
        orc_anim_interpolate(&g_anims[index], g_ray_time, m, mi);
        memcpy(out12, mi, 48);
    } else {
        memcpy(out12, in->m_inv, 48);
    }
}

/* ---- primitive dispatch (cpu/primitive.cpp:24-32 -> shapes.cpp:320-358, 1131-1156) --- */
static inline int prim_test(const orc_prim *p, const float *verts, const float o[3],
                            const float d[3], float tmax, float res[4]) {
    if (p->kind == 0 || (p->kind >= 4 && p->kind <= 7)) {
        return orc_triangle(o, d, tmax, verts + 3 * (size_t)p->v[0], verts + 3 * (size_t)p->v[1],
                            verts + 3 * (size_t)p->v[2], res);
    } else {
        float uvt[3];
        if (!orc_bilinear_patch(o, d, tmax, verts + 3 * (size_t)p->v[0],
                                verts + 3 * (size_t)p->v[1], verts + 3 * (size_t)p->v[2],
                                verts + 3 * (size_t)p->v[3], uvt))
            return 0;
        res[0] = uvt[0];
        res[1] = uvt[1];
        res[2] = 0.0f;
        res[3] = uvt[2];
        return 1;
    }
}

/* ---- GeometricPrimitive::Intersect with a constant alpha, cpu/primitive.cpp:50-77 ------------- */
/* prim kinds 4 / 5: a triangle (mesh without per-vertex normals; 5 = flipped orientation) behind a
 * GeometricPrimitive whose alpha texture evaluates to the constant in v[3] (float bit pattern);
 * kinds 6 / 7: the same for a mesh WITH per-vertex shading normals (orc_set_vertex_normals): the interaction
 * the re-traced ray is spawned from then carries n = FaceForward(n, ns) (shapes.h:939-951).
 * Every Triangle::Intersect call counts in nTriTests, the recursive ones too. */
int orc_triangle_interaction(const float p9[9], const float *uv6, const float *n9, const float *s9,
                             int flip_normal, const float b[3], const float wo[3], float time,
                             int face_index, float out[44]);
uint64_t orc_hash_6f(const float a[3], const float b[3]);
float orc_hash_float_6f(const float a[3], const float b[3]);
void orc_offset_ray_origin(const float lo[3], const float hi[3], const float n[3], const float w[3], float po[3]);

/* The re-trace after a rejected hit (:63-69) is the reference's recursion unrolled ONCE: for a planar
 * triangle the ray spawned off its own surface cannot hit it again, and the recursion ends there.  Only
 * degenerate rays (NaN / zero direction, NaN tHit) make the second test succeed, and then the reference
 * recurses without bound; the library's contract for that case is "the record is void, the caller
 * re-traces the ray" (*host_io = 1), and this restatement follows it instead of overflowing the stack. */
static const float *g_vertex_normals = NULL; /* 3 floats per vertex, indexed like verts; kinds 6 / 7 read it */
void orc_set_vertex_normals(const float *normals) { g_vertex_normals = normals; }

/* prim kinds 8 .. 15 = 8 + flipped + 2 * smooth + 4 * uv: a BILINEAR PATCH behind such a GeometricPrimitive (flipped
 * orientation; the mesh has per-vertex normals; the mesh has (u, v) coordinates, whose (s, t) reparametrisation the
 * interaction's normal then goes through, shapes.h:1414-1437).  The constant alpha comes from a per-primitive array
 * (orc_set_prim_alpha: the patch needs all four v[]).  A non-planar patch CAN be met again by the ray spawned
 * off its own surface, so the recursion of :63-69 is real here: it is followed for up to ORC_ALPHA_PATCH_DEPTH
 * re-traces (a doubly ruled quadric meets a line twice at most; a third re-trace that still hits is numerical
 * self-intersection), beyond which the record is void and the caller re-traces the ray (*host_io = 1).
 * siNext->tHit += si->tHit (:67-68) unwinds from the deepest level: ((t_k + t_{k-1}) + ...) + t_0. */
#define ORC_ALPHA_PATCH_DEPTH 3
static const float *g_vertex_uvs = NULL; /* 2 floats per vertex; kinds 12 .. 15 (8 + flipped + 2 smooth + 4 uv) read it */
void orc_set_vertex_uvs(const float *uvs) { g_vertex_uvs = uvs; }
static const float *g_prim_alpha = NULL;
static const orc_prim *g_prim_alpha_base = NULL;
void orc_set_prim_alpha(const float *alpha) { g_prim_alpha = alpha; }
int orc_patch_interaction(const float p12[12], const float *uv8, const float *n12, int flip_normal,
                          const float hit_uv[2], const float wo[3], float time, int face_index, float out[50]);

static int alpha_patch_intersect(const orc_prim *p, const float *verts, const float o[3], const float d[3],
                                 float tmax, float res[4], int *tests, int *host_io) {
    const float a = g_prim_alpha ? g_prim_alpha[p - g_prim_alpha_base] : 1.0f;
    const int smooth = ((p->kind - 8) & 2) && g_vertex_normals;
    const int flip = (p->kind - 8) & 1;
    const int has_uv = ((p->kind - 8) & 4) && g_vertex_uvs;
    float oc[3] = {o[0], o[1], o[2]}, tm = tmax, ts[ORC_ALPHA_PATCH_DEPTH];
    int k = 0;
    for (;;) {
        float r[4];
        ++*tests;
        if (!prim_test(p, verts, oc, d, tm, r)) return 0; /* :52-54 at this level: the whole chain returns {} */
        int rejected = 0;
        if (a < 1) { /* :58 */
            const float u = (a <= 0) ? 1.f : orc_hash_float_6f(oc, d); /* :60, the ray of THIS level */
            rejected = u > a;
        }
        if (!rejected) {
            float total = r[3];
            for (int j = k - 1; j >= 0; --j) total = total + ts[j]; /* :67-68, innermost first */
            res[0] = r[0], res[1] = r[1], res[2] = 0.0f, res[3] = total;
            return 1;
        }
        if (k == ORC_ALPHA_PATCH_DEPTH) {
            *host_io = 1;
            return 0;
        }
        ts[k++] = r[3];
        float p12[12], n12[12], uv8[8], wo[3] = {-d[0], -d[1], -d[2]}, rec[50], on[3];
        for (int j = 0; j < 4; ++j) memcpy(p12 + 3 * j, verts + 3 * (size_t)p->v[j], 12);
        if (smooth)
            for (int j = 0; j < 4; ++j) memcpy(n12 + 3 * j, g_vertex_normals + 3 * (size_t)p->v[j], 12);
        if (has_uv)
            for (int j = 0; j < 4; ++j) memcpy(uv8 + 2 * j, g_vertex_uvs + 2 * (size_t)p->v[j], 8);
        orc_patch_interaction(p12, has_uv ? uv8 : NULL, smooth ? n12 : NULL, flip, r, wo, 0.0f, 0, rec);
        orc_offset_ray_origin(rec + 38, rec + 41, rec + 11, d, on); /* rNext = si->intr.SpawnRay(r.d) */
        memcpy(oc, on, 12);
        tm = tm - r[3]; /* Intersect(rNext, tMax - si->tHit) */
    }
}

static int alpha_intersect(const orc_prim *p, const float *verts, const float o[3], const float d[3],
                           float tmax, float res[4], int *tests, int *host_io) {
    if (p->kind >= 8 && p->kind <= 15) return alpha_patch_intersect(p, verts, o, d, tmax, res, tests, host_io);
    float r[4];
    ++*tests;
    if (!prim_test(p, verts, o, d, tmax, r)) return 0; /* :52-54 */
    float a;
    memcpy(&a, &p->v[3], 4);
    if (a < 1) { /* :58 */
        const float u = (a <= 0) ? 1.f : orc_hash_float_6f(o, d); /* :60 */
        if (u > a) {
            /* :63-69 ignore this intersection and trace a new ray: rNext = si->intr.SpawnRay(r.d) */
            float p9[9], n9[9], wo[3] = {-d[0], -d[1], -d[2]}, rec[44], on[3], rn[4];
            const int smooth = (p->kind == 6 || p->kind == 7) && g_vertex_normals;
            memcpy(p9, verts + 3 * (size_t)p->v[0], 12);
            memcpy(p9 + 3, verts + 3 * (size_t)p->v[1], 12);
            memcpy(p9 + 6, verts + 3 * (size_t)p->v[2], 12);
            if (smooth)
                for (int j = 0; j < 3; ++j) memcpy(n9 + 3 * j, g_vertex_normals + 3 * (size_t)p->v[j], 12);
            orc_triangle_interaction(p9, NULL, smooth ? n9 : NULL, NULL, p->kind == 5 || p->kind == 7, r, wo, 0.0f, 0, rec);
            orc_offset_ray_origin(rec + 38, rec + 41, rec + 11, d, on);
            ++*tests; /* Triangle::Intersect counts the re-test too */
            if (prim_test(p, verts, on, d, tmax - r[3], rn)) *host_io = 1;
            return 0;
        }
    }
    memcpy(res, r, 16);
    return 1;
}

/* ---- BVHAggregate::Intersect, aggregates.cpp:529-579 -------------------------------- */
/* One BVHAggregate::Intersect over the tree rooted at `root`; hit/tmax/counters are carried
 * by the caller so that a TransformedPrimitive's child aggregate (cpu/primitive.cpp:112-126)
 * adds to the same bvhNodesVisited / nTriTests as the reference's global counters do. */
static void closest_tree(const orc_node *nodes, const orc_prim *prims, const float *verts,
                         const orc_instance *instances, int root, const float o[3],
                         const float d[3], float *tmax_io, orc_hit *hit, int *visited_io,
                         int *tests_io, int instance_tag, int *host_io) {
    float tmax = *tmax_io;
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    int neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
    int to_visit = 0, cur = root, visited = *visited_io, tests = *tests_io;
    int stack[64];
    for (;;) {
        ++visited;
        const orc_node *nd = &nodes[cur];
        if (slab_test(nd->pmin, nd->pmax, o, tmax, inv, neg)) {
            if (nd->nprims > 0) {
                for (int i = 0; i < nd->nprims; ++i) {
                    const orc_prim *p = &prims[nd->offset + i];
                    float r[4];
                    if (p->kind == 3) { /* host-only primitive: the ray's record is void */
                        *host_io = 1;
                        continue;
                    }
                    if (p->kind == 2) { /* TransformedPrimitive::Intersect */
                        const orc_instance *in = &instances[p->v[0]];
                        float x[7], mi12[12];
                        instance_minv(in, p->v[0], mi12); /* AnimatedPrimitive: Interpolate(r.time), primitive.cpp:143 */
                        orc_apply_inverse_ray(mi12, o, d, tmax, x);
                        float inner_tmax = x[6];
                        orc_hit inner = *hit;
                        inner.prim = -1;
                        closest_tree(nodes, prims, verts, instances, in->root, x, x + 3,
                                     &inner_tmax, &inner, &visited, &tests, p->v[0] + 1, host_io);
                        if (inner.prim >= 0) { /* si = primSi; tMax = si->tHit */
                            *hit = inner;
                            tmax = inner.t;
                        }
                        continue;
                    }
                    int primHit;
                    if (p->kind >= 4 && p->kind <= 15) {
                        primHit = alpha_intersect(p, verts, o, d, tmax, r, &tests, host_io);
                    } else {
                        ++tests;
                        primHit = prim_test(p, verts, o, d, tmax, r);
                    }
                    if (primHit) {
                        hit->prim = p->id;
                        hit->b0 = r[0];
                        hit->b1 = r[1];
                        hit->b2 = r[2];
                        hit->t = r[3];
                        hit->instance = instance_tag;
                        tmax = r[3];
                    }
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                if (neg[nd->axis]) {
                    stack[to_visit++] = cur + 1;
                    cur = nd->offset;
                } else {
                    stack[to_visit++] = nd->offset;
                    cur = cur + 1;
                }
            }
        } else {
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        }
    }
    *tmax_io = tmax;
    *visited_io = visited;
    *tests_io = tests;
}

static void closest_one(const orc_node *nodes, const orc_prim *prims, const float *verts,
                        const orc_instance *instances, const orc_ray *ray, orc_hit *hit) {
    float tmax = ray->tmax;
    int visited = 0, tests = 0;
    hit->prim = -1;
    hit->t = tmax;
    hit->b0 = hit->b1 = hit->b2 = 0.0f;
    hit->instance = 0;
    int host = 0;
    closest_tree(nodes, prims, verts, instances, 0, ray->o, ray->d, &tmax, hit, &visited, &tests, 0,
                 &host);
    if (hit->prim < 0) hit->t = ray->tmax;
    if (host) hit->instance = -1; /* reached a host-only primitive: record void (nnbvh.h) */
    hit->nodes_visited = visited;
    hit->prim_tests = tests;
}

/* ---- BVHAggregate::IntersectP, aggregates.cpp:581-624 ------------------------------- */
static int any_tree(const orc_node *nodes, const orc_prim *prims, const float *verts,
                    const orc_instance *instances, int root, const float o[3], const float d[3],
                    float tmax, int *visited_io, int *tests_io, int *host_io) {
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    int neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
    int to_visit = 0, cur = root, visited = *visited_io, tests = *tests_io, found = 0;
    int stack[64];
    for (;;) {
        ++visited;
        const orc_node *nd = &nodes[cur];
        if (slab_test(nd->pmin, nd->pmax, o, tmax, inv, neg)) {
            if (nd->nprims > 0) {
                for (int i = 0; i < nd->nprims; ++i) {
                    const orc_prim *p = &prims[nd->offset + i];
                    float r[4];
                    if (p->kind == 3) {
                        *host_io = 1;
                        continue;
                    }
                    if (p->kind == 2) { /* TransformedPrimitive::IntersectP, primitive.cpp:128-131 */
                        const orc_instance *in = &instances[p->v[0]];
                        float x[7], mi12[12];
                        instance_minv(in, p->v[0], mi12);
                        orc_apply_inverse_ray(mi12, o, d, tmax, x);
                        if (any_tree(nodes, prims, verts, instances, in->root, x, x + 3, x[6],
                                     &visited, &tests, host_io)) {
                            found = 1;
                            goto done;
                        }
                        continue;
                    }
                    int primHit; /* GeometricPrimitive::IntersectP with alpha = Intersect(...).has_value(), :79-81 */
                    if (p->kind >= 4 && p->kind <= 15) {
                        primHit = alpha_intersect(p, verts, o, d, tmax, r, &tests, host_io);
                    } else {
                        ++tests;
                        primHit = prim_test(p, verts, o, d, tmax, r);
                    }
                    if (primHit) {
                        found = 1;
                        goto done;
                    }
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                if (neg[nd->axis]) {
                    stack[to_visit++] = cur + 1;
                    cur = nd->offset;
                } else {
                    stack[to_visit++] = nd->offset;
                    cur = cur + 1;
                }
            }
        } else {
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        }
    }
done:
    *visited_io = visited;
    *tests_io = tests;
    return found;
}

static int any_one(const orc_node *nodes, const orc_prim *prims, const float *verts,
                   const orc_instance *instances, const orc_ray *ray, int *visited_out,
                   int *tests_out) {
    int visited = 0, tests = 0, host = 0;
    int found = any_tree(nodes, prims, verts, instances, 0, ray->o, ray->d, ray->tmax, &visited,
                         &tests, &host);
    *visited_out = visited;
    *tests_out = tests;
    return found ? 1 : (host ? 2 : 0); /* 2: unknown, a host-only primitive was reached */
}

/* ---- batch drivers (pthread fan-out mirrors ParallelFor chunking, util/parallel.cpp:291-299) */
typedef struct {
    const orc_node *nodes;
    const orc_prim *prims;
    const float *verts;
    const orc_instance *instances;
    const orc_anim *anims;
    const orc_ray *rays;
    int64_t begin, end;
    orc_hit *hits;
    uint8_t *occ;
    int32_t *visited;
    int32_t *tests;
    int any;
} orc_job;

static void *orc_worker(void *arg) {
    orc_job *j = (orc_job *)arg;
    g_anims = j->anims;
    for (int64_t i = j->begin; i < j->end; ++i) {
        g_ray_time = j->rays[i].time;
        if (j->any) {
            int v, t;
            int f = any_one(j->nodes, j->prims, j->verts, j->instances, &j->rays[i], &v, &t);
            j->occ[i] = (uint8_t)f;
            if (j->visited) j->visited[i] = v;
            if (j->tests) j->tests[i] = t;
        } else {
            closest_one(j->nodes, j->prims, j->verts, j->instances, &j->rays[i], &j->hits[i]);
        }
    }
    return NULL;
}

static void orc_run(orc_job base, int64_t n, int nthreads) {
    if (nthreads <= 1 || n < 2 * (int64_t)nthreads) {
        base.begin = 0;
        base.end = n;
        orc_worker(&base);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    orc_job *jobs = (orc_job *)malloc(sizeof(orc_job) * (size_t)nthreads);
    for (int k = 0; k < nthreads; ++k) {
        jobs[k] = base;
        jobs[k].begin = n * k / nthreads;
        jobs[k].end = n * (k + 1) / nthreads;
        pthread_create(&th[k], NULL, orc_worker, &jobs[k]);
    }
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    free(jobs);
    free(th);
}

void orc_intersect_closest(const orc_node *nodes, int n_nodes, const orc_prim *prims,
                           const float *verts, const orc_ray *rays, int64_t n, orc_hit *hits,
                           int nthreads) {
    (void)n_nodes;
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    orc_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prims = prims;
    j.verts = verts;
    j.rays = rays;
    j.hits = hits;
    j.any = 0;
    orc_run(j, n, nthreads);
}

void orc_intersect_any(const orc_node *nodes, int n_nodes, const orc_prim *prims,
                       const float *verts, const orc_ray *rays, int64_t n, uint8_t *occluded,
                       int32_t *nodes_visited, int32_t *prim_tests, int nthreads) {
    (void)n_nodes;
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    orc_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prims = prims;
    j.verts = verts;
    j.rays = rays;
    j.occ = occluded;
    j.visited = nodes_visited;
    j.tests = prim_tests;
    j.any = 1;
    orc_run(j, n, nthreads);
}

void orc_intersect_closest_inst(const orc_node *nodes, const orc_prim *prims, const float *verts,
                                const orc_instance *instances, const orc_ray *rays, int64_t n,
                                orc_hit *hits, int nthreads) {
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    orc_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prims = prims;
    j.verts = verts;
    j.instances = instances;
    j.rays = rays;
    j.hits = hits;
    orc_run(j, n, nthreads);
}

void orc_intersect_any_inst(const orc_node *nodes, const orc_prim *prims, const float *verts,
                            const orc_instance *instances, const orc_ray *rays, int64_t n,
                            uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests,
                            int nthreads) {
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    orc_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prims = prims;
    j.verts = verts;
    j.instances = instances;
    j.rays = rays;
    j.occ = occluded;
    j.visited = nodes_visited;
    j.tests = prim_tests;
    j.any = 1;
    orc_run(j, n, nthreads);
}

void orc_intersect_closest_anim(const orc_node *nodes, const orc_prim *prims, const float *verts,
                                const orc_instance *instances, const orc_anim *anims, const orc_ray *rays,
                                int64_t n, orc_hit *hits, int nthreads) {
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    orc_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prims = prims;
    j.verts = verts;
    j.instances = instances;
    j.anims = anims;
    j.rays = rays;
    j.hits = hits;
    orc_run(j, n, nthreads);
}

void orc_intersect_any_anim(const orc_node *nodes, const orc_prim *prims, const float *verts,
                            const orc_instance *instances, const orc_anim *anims, const orc_ray *rays,
                            int64_t n, uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests,
                            int nthreads) {
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    orc_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prims = prims;
    j.verts = verts;
    j.instances = instances;
    j.anims = anims;
    j.rays = rays;
    j.occ = occluded;
    j.visited = nodes_visited;
    j.tests = prim_tests;
    j.any = 1;
    orc_run(j, n, nthreads);
}

void orc_brute_closest(const orc_prim *prims, int n_prims, const float *verts,
                       const orc_ray *rays, int64_t n, orc_hit *hits) {
    for (int64_t i = 0; i < n; ++i) {
        const orc_ray *ray = &rays[i];
        orc_hit *h = &hits[i];
        float tmax = ray->tmax;
        h->prim = -1;
        h->t = tmax;
        h->b0 = h->b1 = h->b2 = 0.0f;
        h->nodes_visited = 0;
        h->instance = 0;
        for (int k = 0; k < n_prims; ++k) {
            float r[4];
            if (prim_test(&prims[k], verts, ray->o, ray->d, tmax, r)) {
                h->prim = prims[k].id;
                h->b0 = r[0];
                h->b1 = r[1];
                h->b2 = r[2];
                h->t = r[3];
                tmax = r[3];
            }
        }
        h->prim_tests = n_prims;
    }
}

/* ------------------------------------------------------------------------------------------
 * Wavefront queue rules, restated on work-item indices.
 *
 * EnqueueWorkAfterMiss (wavefront/intersect.h:16-30): a missing ray with a medium goes to the
 * medium-sample queue, otherwise to the escaped-ray queue.
 * EnqueueWorkAfterIntersection (intersect.h:49-156): a hit ray with a medium goes to the
 * medium-sample queue and nothing else (:56-90); otherwise a surface without material is an
 * interface and the ray continues via nextRayQueue (:103-111); otherwise an emissive surface is
 * also pushed to the hit-area-light queue (:113-120) and the item goes to the basic or the
 * universal material-evaluation queue (:122-128).  cls bits: 1 = universal, 2 = interface,
 * 4 = area light.  Queue order here is index order; the reference's is whatever its threads
 * produce, so tests compare queue CONTENTS.
 * queues / sizes order: escaped, hit_area_light, basic, universal, medium_sample, next_ray. */
void orc_wavefront_enqueue_closest(const orc_hit *hits, int n, const uint8_t *has_medium,
                                   const uint8_t *prim_class, int64_t n_class,
                                   int32_t *const queues[6], int32_t sizes[6]) {
    for (int k = 0; k < 6; ++k) sizes[k] = 0;
    for (int i = 0; i < n; ++i) {
        const int medium = has_medium && has_medium[i];
        if (hits[i].prim < 0) {
            const int q = medium ? 4 : 0;
            queues[q][sizes[q]++] = i;
            continue;
        }
        if (medium) {
            queues[4][sizes[4]++] = i;
            continue;
        }
        unsigned cls = 0;
        if (prim_class && hits[i].prim < n_class) cls = prim_class[hits[i].prim];
        if (cls & 2u) {
            queues[5][sizes[5]++] = i;
            continue;
        }
        if (cls & 4u) queues[1][sizes[1]++] = i;
        const int q = (cls & 1u) ? 3 : 2;
        queues[q][sizes[q]++] = i;
    }
}

/* RecordShadowRayResult (wavefront/intersect.h:32-47) with SampledSpectrum = 4 floats
 * (util/spectrum.h:36): Ld / (r_u + r_l).Average(), Average() = left-to-right sum / 4
 * (spectrum.h:256-261), operator/(Float) divides each component (spectrum.h:163-174). */
void orc_record_shadow(const uint8_t *occluded, int n, const float *Ld, const float *r_u,
                       const float *r_l, const int32_t *pixel_index, float *L) {
    for (int i = 0; i < n; ++i) {
        if (occluded[i]) continue;
        float s[4];
        for (int c = 0; c < 4; ++c) s[c] = r_u[4 * i + c] + r_l[4 * i + c];
        float sum = s[0];
        for (int c = 1; c < 4; ++c) sum += s[c];
        const float avg = sum / 4;
        float *dst = L + 4 * (int64_t)pixel_index[i];
        for (int c = 0; c < 4; ++c) {
            const float ld = Ld[4 * i + c] / avg;
            dst[c] = dst[c] + ld;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Triangle::InteractionFromIntersection (shapes.h:884-1010) + the SurfaceInteraction constructor
 * and SetShadingGeometry it runs (interaction.h:164-214).  Output record (44 floats):
 *   [0..2] p() = interval midpoints, [3..5] pi.Error(), [6..7] uv, [8..10] wo, [11..13] n,
 *   [14..16] dpdu, [17..19] dpdv, [20..22] shading.n, [23..25] shading.dpdu, [26..28] shading.dpdv,
 *   [29..31] shading.dndu, [32..34] shading.dndv, [35] time, [36] faceIndex, [37] 0,
 *   [38..40] pi lower bounds, [41..43] pi upper bounds.
 * uv6 / n9 / s9 may be NULL (mesh without that attribute).  Returns 0 when the reference would
 * CHECK-abort (zero-area triangle, which IntersectTriangle never reports as hit). */
static inline void orc_dop_v(float a, const float b[3], float c, const float d[3], float out[3]) {
    /* DifferenceOfProducts(Float, Vector3f, Float, Vector3f): math.h:569-575 with the
     * component-wise vector FMA of vecmath.h:415-417 */
    for (int k = 0; k < 3; ++k) {
        float cd = c * d[k];
        float diff = fmaf(a, b[k], -cd);
        float err = fmaf(-c, d[k], cd);
        out[k] = diff + err;
    }
}
static inline float orc_sop(float a, float b, float c, float d) { /* SumOfProducts, math.h:577-583 */
    float cd = c * d;
    float s = fmaf(a, b, cd);
    float err = fmaf(c, d, -cd);
    return s + err;
}
static inline float orc_dot_n(const float n[3], const float v[3]) { /* Dot(Normal3, .), vecmath.h:1056-1075 */
    return fmaf(n[0], v[0], orc_sop(n[1], v[1], n[2], v[2]));
}
static inline void orc_normalize(const float v[3], float out[3]) { /* v / Length(v), vecmath.h:953-961 */
    float len = sqrtf(orc_len2(v));
    for (int k = 0; k < 3; ++k) out[k] = v[k] / len;
}
static inline void orc_coordinate_system(const float v1[3], float v2[3], float v3[3]) { /* vecmath.h:1007-1013 */
    float sign = copysignf(1.0f, v1[2]);
    float a = -1 / (sign + v1[2]);
    float b = v1[0] * v1[1] * a;
    v2[0] = 1 + sign * (v1[0] * v1[0]) * a;
    v2[1] = sign * b;
    v2[2] = -sign * v1[0];
    v3[0] = b;
    v3[1] = sign + (v1[1] * v1[1]) * a;
    v3[2] = -v1[1];
}
static inline void orc_bary3(const float b[3], const float *a0, const float *a1, const float *a2, int dim,
                             float *out) { /* b0 * a0 + b1 * a1 + b2 * a2, left to right */
    for (int k = 0; k < dim; ++k) out[k] = (b[0] * a0[k] + b[1] * a1[k]) + b[2] * a2[k];
}

/* how often each rare branch was taken (tests check that the vectors reach them):
 * 0 degenerate uv, 1 dpdu x dpdv == 0, 2 zero interpolated normal, 3 zero tangent, 4 tangent
 * parallel to normal, 5 degenerate uv in the dndu/dndv block, 6 equal normals (dn == 0),
 * 7 shading dpdu rescaled, 8 geometric normal flipped to the shading side */
long orc_interaction_branches[9];

int orc_triangle_interaction(const float p9[9], const float *uv6, const float *n9, const float *s9,
                             int flip_normal, const float b[3], const float wo[3], float time,
                             int face_index, float out[44]) {
    const float *p0 = p9, *p1 = p9 + 3, *p2 = p9 + 6;
    static const float uv_default[6] = {0, 0, 1, 0, 1, 1};
    const float *uv = uv6 ? uv6 : uv_default;
    float duv02[2] = {uv[0] - uv[4], uv[1] - uv[5]}, duv12[2] = {uv[2] - uv[4], uv[3] - uv[5]};
    float dp02[3], dp12[3];
    for (int k = 0; k < 3; ++k) {
        dp02[k] = p0[k] - p2[k];
        dp12[k] = p1[k] - p2[k];
    }
    float determinant = orc_dop(duv02[0], duv12[1], duv02[1], duv12[0]);
    float dpdu[3] = {0, 0, 0}, dpdv[3] = {0, 0, 0};
    int degenerate_uv = fabsf(determinant) < 1e-9f;
    if (!degenerate_uv) {
        float invdet = 1 / determinant, t[3];
        orc_dop_v(duv12[1], dp02, duv02[1], dp12, t);
        for (int k = 0; k < 3; ++k) dpdu[k] = invdet * t[k];
        orc_dop_v(duv02[0], dp12, duv12[0], dp02, t);
        for (int k = 0; k < 3; ++k) dpdv[k] = invdet * t[k];
    }
    float c[3];
    orc_cross(dpdu, dpdv, c);
    if (degenerate_uv) ++orc_interaction_branches[0];
    else if (orc_len2(c) == 0) ++orc_interaction_branches[1];
    if (degenerate_uv || orc_len2(c) == 0) {
        float e20[3], e10[3], ng[3], ngn[3];
        for (int k = 0; k < 3; ++k) {
            e20[k] = p2[k] - p0[k];
            e10[k] = p1[k] - p0[k];
        }
        orc_cross(e20, e10, ng);
        if (orc_len2(ng) == 0) { /* shapes.h:916-919: redo the cross product in double */
            double v[3] = {e20[0], e20[1], e20[2]}, w[3] = {e10[0], e10[1], e10[2]}, r[3];
            for (int k = 0; k < 3; ++k) {
                int i1 = (k + 1) % 3, i2 = (k + 2) % 3;
                double cd = v[i2] * w[i1];
                double diff = fma(v[i1], w[i2], -cd);
                double err = fma(-v[i2], w[i1], cd);
                r[k] = diff + err;
            }
            for (int k = 0; k < 3; ++k) ng[k] = (float)r[k];
            if (orc_len2(ng) == 0) return 0;
        }
        orc_normalize(ng, ngn);
        orc_coordinate_system(ngn, dpdu, dpdv);
    }
    float p_hit[3], uv_hit[2], p_err[3];
    orc_bary3(b, p0, p1, p2, 3, p_hit);
    orc_bary3(b, uv, uv + 2, uv + 4, 2, uv_hit);
    for (int k = 0; k < 3; ++k) {
        float s = (fabsf(b[0] * p0[k]) + fabsf(b[1] * p1[k])) + fabsf(b[2] * p2[k]);
        p_err[k] = orc_gamma(7) * s;
    }
    /* isect.n = isect.shading.n = Normalize(Cross(dp02, dp12)), flipped by orientation (:933-936);
     * the constructor's own n (from dpdu x dpdv) is overwritten */
    float n[3], ns[3], t[3];
    orc_cross(dp02, dp12, t);
    orc_normalize(t, n);
    if (flip_normal)
        for (int k = 0; k < 3; ++k) n[k] = -n[k];
    float sdpdu[3], sdpdv[3], dndu[3] = {0, 0, 0}, dndv[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        ns[k] = n[k];
        sdpdu[k] = dpdu[k];
        sdpdv[k] = dpdv[k];
    }
    if (n9 || s9) {
        float nsv[3], ss[3], ts[3];
        if (n9) {
            orc_bary3(b, n9, n9 + 3, n9 + 6, 3, t);
            if (orc_len2(t) > 0) orc_normalize(t, nsv);
            else {
                memcpy(nsv, n, 12);
                ++orc_interaction_branches[2];
            }
        } else
            memcpy(nsv, n, 12);
        if (s9) {
            orc_bary3(b, s9, s9 + 3, s9 + 6, 3, ss);
            if (orc_len2(ss) == 0) {
                memcpy(ss, dpdu, 12);
                ++orc_interaction_branches[3];
            }
        } else
            memcpy(ss, dpdu, 12);
        orc_cross(nsv, ss, ts);
        if (orc_len2(ts) > 0) orc_cross(ts, nsv, ss);
        else {
            orc_coordinate_system(nsv, ss, ts);
            ++orc_interaction_branches[4];
        }
        if (n9) {
            float dn1[3], dn2[3];
            for (int k = 0; k < 3; ++k) {
                dn1[k] = n9[k] - n9[6 + k];
                dn2[k] = n9[3 + k] - n9[6 + k];
            }
            float det2 = orc_dop(duv02[0], duv12[1], duv02[1], duv12[0]);
            int degenerate2 = (double)fabsf(det2) < 1e-9; /* :963: a double literal here */
            if (degenerate2) {
                float a[3], bb[3], dn[3];
                ++orc_interaction_branches[5];
                for (int k = 0; k < 3; ++k) {
                    a[k] = n9[6 + k] - n9[k];
                    bb[k] = n9[3 + k] - n9[k];
                }
                orc_cross(a, bb, dn);
                if (orc_len2(dn) != 0) orc_coordinate_system(dn, dndu, dndv);
                else ++orc_interaction_branches[6];
            } else {
                float inv_det = 1 / det2;
                orc_dop_v(duv12[1], dn1, duv02[1], dn2, t);
                for (int k = 0; k < 3; ++k) dndu[k] = inv_det * t[k];
                orc_dop_v(duv02[0], dn2, duv12[0], dn1, t);
                for (int k = 0; k < 3; ++k) dndv[k] = inv_det * t[k];
            }
        }
        /* SetShadingGeometry(ns, ss, ts, dndu, dndv, true): interaction.h:194-214 */
        memcpy(ns, nsv, 12);
        if (orc_dot_n(n, ns) < 0.f) {
            for (int k = 0; k < 3; ++k) n[k] = -n[k];
            ++orc_interaction_branches[8];
        }
        memcpy(sdpdu, ss, 12);
        memcpy(sdpdv, ts, 12);
        while (orc_len2(sdpdu) > 1e16f || orc_len2(sdpdv) > 1e16f) {
            ++orc_interaction_branches[7];
            for (int k = 0; k < 3; ++k) {
                sdpdu[k] /= 1e8f;
                sdpdv[k] /= 1e8f;
            }
        }
    }
    for (int k = 0; k < 3; ++k) {
        orc_ivl iv = ivl_from_value_and_error(p_hit[k], p_err[k]);
        out[k] = (iv.lo + iv.hi) / 2;
        out[3 + k] = (iv.hi - iv.lo) / 2;
        out[38 + k] = iv.lo;
        out[41 + k] = iv.hi;
        out[11 + k] = n[k];
        out[14 + k] = dpdu[k];
        out[17 + k] = dpdv[k];
        out[20 + k] = ns[k];
        out[23 + k] = sdpdu[k];
        out[26 + k] = sdpdv[k];
        out[29 + k] = dndu[k];
        out[32 + k] = dndv[k];
    }
    orc_normalize(wo, out + 8); /* Interaction(): wo(Normalize(wo)), interaction.h:32-33 */
    out[6] = uv_hit[0];
    out[7] = uv_hit[1];
    out[35] = time;
    out[36] = (float)face_index;
    out[37] = 0;
    return 1;
}

/* records as oracle/ref_interaction.cpp reads them (45 floats, see there).  That harness builds its
 * mesh with the reference's TriangleMesh constructor, which stores the normals NEGATED when
 * reverseOrientation is set (util/mesh.cpp:52-58); the same is done to the input here, so that
 * orc_triangle_interaction sees mesh->n as InteractionFromIntersection does. */
void orc_triangle_interaction_batch(const float *in45, int n, float *out44) {
    for (int i = 0; i < n; ++i) {
        const float *r = in45 + 45 * (size_t)i;
        int flags = (int)r[19];
        float nn[9];
        for (int k = 0; k < 9; ++k) nn[k] = (flags & 8) ? -r[26 + k] : r[26 + k];
        orc_triangle_interaction(r, (flags & 1) ? r + 20 : NULL, (flags & 2) ? nn : NULL,
                                 (flags & 4) ? r + 36 : NULL, (flags & 8) != 0, r + 9, r + 12, r[18], 7 + i,
                                 out44 + 44 * (size_t)i);
    }
}

/* ------------------------------------------------------------------------------------------
 * BilinearPatch::InteractionFromIntersection (shapes.h:1396-1489) + RotateFromTo
 * (util/transform.h:249-270) + the SurfaceInteraction constructor / SetShadingGeometry.
 * p12 = p00 p10 p01 p11; uv8 / n12 in the same vertex order, NULL = mesh without that attribute;
 * hit_uv = the (u, v) of BilinearIntersection.  Output record (50 floats): [0..43] as
 * orc_triangle_interaction, [44..46] geometric dndu, [47..49] geometric dndv. */
static inline void orc_lerp_n(float t, const float *a, const float *b, int dim, float *out) {
    /* Lerp(t, a, b) = (1 - t) * a + t * b  (vecmath.h:203-205, 410-412) */
    const float s = 1 - t;
    for (int k = 0; k < dim; ++k) out[k] = s * a[k] + t * b[k];
}
static inline void orc_scale_add2(const float a[3], float sa, const float b[3], float sb, float out[3]) {
    /* a * sa + b * sb with Tuple3::operator*(U s) = {s * x, ...} (vecmath.h:350-353) */
    for (int k = 0; k < 3; ++k) out[k] = sa * a[k] + sb * b[k];
}

/* rare-branch counters of orc_patch_interaction: 0 (s,t) derivatives adopted, 1 dpdt negated, 2 a
 * 1e-8 derivative test zeroed a factor, 3 zero interpolated normal, 4 / 5 reflection about y / z in
 * RotateFromTo, 6 geometric normal flipped to the shading side, 7 cross(dpds, dpdt) == 0 */
long orc_patch_branches[8];

int orc_patch_interaction(const float p12[12], const float *uv8, const float *n12, int flip_normal,
                          const float hit_uv[2], const float wo[3], float time, int face_index,
                          float out[50]) {
    const float *p00 = p12, *p10 = p12 + 3, *p01 = p12 + 6, *p11 = p12 + 9;
    const float u = hit_uv[0], v = hit_uv[1];
    float a[3], b[3], p[3], dpdu[3], dpdv[3];
    orc_lerp_n(v, p00, p01, 3, a);
    orc_lerp_n(v, p10, p11, 3, b);
    orc_lerp_n(u, a, b, 3, p);
    for (int k = 0; k < 3; ++k) dpdu[k] = b[k] - a[k];
    orc_lerp_n(u, p01, p11, 3, a);
    orc_lerp_n(u, p00, p10, 3, b);
    for (int k = 0; k < 3; ++k) dpdv[k] = a[k] - b[k];
    float st[2] = {u, v};
    float duds = 1, dudt = 0, dvds = 0, dvdt = 1;
    if (uv8) {
        const float *uv00 = uv8, *uv10 = uv8 + 2, *uv01 = uv8 + 4, *uv11 = uv8 + 6;
        float s0[2], s1[2], dstdu[2], dstdv[2];
        orc_lerp_n(v, uv00, uv01, 2, s0);
        orc_lerp_n(v, uv10, uv11, 2, s1);
        orc_lerp_n(u, s0, s1, 2, st);
        for (int k = 0; k < 2; ++k) dstdu[k] = s1[k] - s0[k];
        orc_lerp_n(u, uv01, uv11, 2, s0);
        orc_lerp_n(u, uv00, uv10, 2, s1);
        for (int k = 0; k < 2; ++k) dstdv[k] = s0[k] - s1[k];
        duds = fabsf(dstdu[0]) < 1e-8f ? 0 : 1 / dstdu[0];
        dvds = fabsf(dstdv[0]) < 1e-8f ? 0 : 1 / dstdv[0];
        dudt = fabsf(dstdu[1]) < 1e-8f ? 0 : 1 / dstdu[1];
        dvdt = fabsf(dstdv[1]) < 1e-8f ? 0 : 1 / dstdv[1];
        if (duds == 0 || dvds == 0 || dudt == 0 || dvdt == 0) ++orc_patch_branches[2];
        float dpds[3], dpdt[3], c1[3], c0[3];
        orc_scale_add2(dpdu, duds, dpdv, dvds, dpds);
        orc_scale_add2(dpdu, dudt, dpdv, dvdt, dpdt);
        orc_cross(dpds, dpdt, c1);
        if (c1[0] != 0 || c1[1] != 0 || c1[2] != 0) {
            orc_cross(dpdu, dpdv, c0);
            ++orc_patch_branches[0];
            if (orc_dot(c0, c1) < 0) {
                for (int k = 0; k < 3; ++k) dpdt[k] = -dpdt[k];
                ++orc_patch_branches[1];
            }
            memcpy(dpdu, dpds, 12);
            memcpy(dpdv, dpdt, 12);
        } else
            ++orc_patch_branches[7];
    }
    /* fundamental forms (:1441-1456); d2Pduu = d2Pdvv = 0 */
    float d2uv[3], zero[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) d2uv[k] = (p00[k] - p01[k]) + (p11[k] - p10[k]);
    const float E = orc_dot(dpdu, dpdu), F = orc_dot(dpdu, dpdv), G = orc_dot(dpdv, dpdv);
    float cr[3], nn[3];
    orc_cross(dpdu, dpdv, cr);
    orc_normalize(cr, nn);
    const float e = orc_dot(nn, zero), f = orc_dot(nn, d2uv), g = orc_dot(nn, zero);
    const float EGF2 = orc_dop(E, G, F, F);
    const float invEGF2 = (EGF2 == 0) ? 0.0f : 1 / EGF2;
    float dndu[3], dndv[3], dnds[3], dndt[3];
    orc_scale_add2(dpdu, (f * F - e * G) * invEGF2, dpdv, (e * F - f * E) * invEGF2, dndu);
    orc_scale_add2(dpdu, (g * F - f * G) * invEGF2, dpdv, (f * F - g * E) * invEGF2, dndv);
    orc_scale_add2(dndu, duds, dndv, dvds, dnds);
    orc_scale_add2(dndu, dudt, dndv, dvdt, dndt);
    memcpy(dndu, dnds, 12);
    memcpy(dndv, dndt, 12);
    float p_err[3];
    for (int k = 0; k < 3; ++k) {
        float s = ((fabsf(p00[k]) + fabsf(p01[k])) + fabsf(p10[k])) + fabsf(p11[k]);
        p_err[k] = orc_gamma(6) * s;
    }
    /* SurfaceInteraction(pi, st, wo, dpdu, dpdv, dndu, dndv, time, flipNormal): interaction.h:164-183 */
    float n[3], ns[3], sdpdu[3], sdpdv[3], sdndu[3], sdndv[3];
    orc_cross(dpdu, dpdv, cr);
    orc_normalize(cr, n);
    if (flip_normal)
        for (int k = 0; k < 3; ++k) n[k] *= -1;
    memcpy(ns, n, 12);
    memcpy(sdpdu, dpdu, 12);
    memcpy(sdpdv, dpdv, 12);
    memcpy(sdndu, dndu, 12);
    memcpy(sdndv, dndv, 12);
    if (n12) {
        const float *n00 = n12, *n10 = n12 + 3, *n01 = n12 + 6, *n11 = n12 + 9;
        float a0[3], a1[3], nsv[3];
        orc_lerp_n(v, n00, n01, 3, a0);
        orc_lerp_n(v, n10, n11, 3, a1);
        orc_lerp_n(u, a0, a1, 3, nsv);
        if (!(orc_len2(nsv) > 0)) ++orc_patch_branches[3];
        if (orc_len2(nsv) > 0) {
            float nsn[3], du[3], dv[3], ds[3], dt[3];
            orc_normalize(nsv, nsn);
            for (int k = 0; k < 3; ++k) du[k] = a1[k] - a0[k];
            orc_lerp_n(u, n01, n11, 3, a0);
            orc_lerp_n(u, n00, n10, 3, a1);
            for (int k = 0; k < 3; ++k) dv[k] = a0[k] - a1[k];
            orc_scale_add2(du, duds, dv, dvds, ds);
            orc_scale_add2(du, dudt, dv, dvdt, dt);
            /* RotateFromTo(Normalize(isect.n), ns): util/transform.h:249-270 */
            float from[3], refl[3] = {0, 0, 0}, uu[3], vv[3], r[3][3];
            orc_normalize(n, from);
            if (fabsf(from[0]) < 0.72f && fabsf(nsn[0]) < 0.72f) refl[0] = 1;
            else if (fabsf(from[1]) < 0.72f && fabsf(nsn[1]) < 0.72f) {
                refl[1] = 1;
                ++orc_patch_branches[4];
            } else {
                refl[2] = 1;
                ++orc_patch_branches[5];
            }
            for (int k = 0; k < 3; ++k) {
                uu[k] = refl[k] - from[k];
                vv[k] = refl[k] - nsn[k];
            }
            const float duu = orc_dot(uu, uu), dvv = orc_dot(vv, vv), duv = orc_dot(uu, vv);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    r[i][j] = ((i == j) ? 1 : 0) - 2 / duu * uu[i] * uu[j] - 2 / dvv * vv[i] * vv[j] +
                              4 * duv / (duu * dvv) * vv[i] * uu[j];
            float rdpdu[3], rdpdv[3];
            for (int i = 0; i < 3; ++i) {
                rdpdu[i] = r[i][0] * dpdu[0] + r[i][1] * dpdu[1] + r[i][2] * dpdu[2];
                rdpdv[i] = r[i][0] * dpdv[0] + r[i][1] * dpdv[1] + r[i][2] * dpdv[2];
            }
            /* SetShadingGeometry(ns, r(dpdu), r(dpdv), dndu, dndv, true) */
            memcpy(ns, nsn, 12);
            if (orc_dot_n(n, ns) < 0.f) {
                for (int k = 0; k < 3; ++k) n[k] = -n[k];
                ++orc_patch_branches[6];
            }
            memcpy(sdpdu, rdpdu, 12);
            memcpy(sdpdv, rdpdv, 12);
            memcpy(sdndu, ds, 12);
            memcpy(sdndv, dt, 12);
            while (orc_len2(sdpdu) > 1e16f || orc_len2(sdpdv) > 1e16f)
                for (int k = 0; k < 3; ++k) {
                    sdpdu[k] /= 1e8f;
                    sdpdv[k] /= 1e8f;
                }
        }
    }
    for (int k = 0; k < 3; ++k) {
        orc_ivl iv = ivl_from_value_and_error(p[k], p_err[k]);
        out[k] = (iv.lo + iv.hi) / 2;
        out[3 + k] = (iv.hi - iv.lo) / 2;
        out[38 + k] = iv.lo;
        out[41 + k] = iv.hi;
        out[11 + k] = n[k];
        out[14 + k] = dpdu[k];
        out[17 + k] = dpdv[k];
        out[20 + k] = ns[k];
        out[23 + k] = sdpdu[k];
        out[26 + k] = sdpdv[k];
        out[29 + k] = sdndu[k];
        out[32 + k] = sdndv[k];
        out[44 + k] = dndu[k];
        out[47 + k] = dndv[k];
    }
    orc_normalize(wo, out + 8);
    out[6] = st[0];
    out[7] = st[1];
    out[35] = time;
    out[36] = (float)face_index;
    out[37] = 0;
    return 1;
}

/* records as oracle/ref_interaction.cpp's "blp" mode reads them (40 floats); normals negated under
 * reverseOrientation like the BilinearPatchMesh constructor does (util/mesh.cpp:216-223) */
void orc_patch_interaction_batch(const float *in40, int n, float *out50) {
    for (int i = 0; i < n; ++i) {
        const float *r = in40 + 40 * (size_t)i;
        int flags = (int)r[18];
        float nn[12];
        for (int k = 0; k < 12; ++k) nn[k] = (flags & 8) ? -r[27 + k] : r[27 + k];
        orc_patch_interaction(r, (flags & 1) ? r + 19 : NULL, (flags & 2) ? nn : NULL, (flags & 8) != 0, r + 12,
                              r + 14, r[17], 7 + i, out50 + 50 * (size_t)i);
    }
}

/* ------------------------------------------------------------------------------------------
 * Transform::operator()(const SurfaceInteraction &) (util/transform.cpp:229-261) for an affine
 * transform: m / m_inv are the 3x4 render-from-primitive matrix and its inverse (row 3 = 0 0 0 1,
 * so the homogeneous weight is exactly 1).  What TransformedPrimitive::Intersect applies to a hit
 * inside an instance (cpu/primitive.cpp:112-125).
 * in39 / out39: pi low[3] high[3], then n, wo, dpdu, dpdv, dndu, dndv, shading n, dpdu, dpdv, dndu,
 * dndv (3 floats each). */
static inline void xf_vec(const float m[12], const float v[3], float out[3]) { /* transform.h:322-326 */
    for (int i = 0; i < 3; ++i) out[i] = m[4 * i] * v[0] + m[4 * i + 1] * v[1] + m[4 * i + 2] * v[2];
}
static inline void xf_normal(const float mi[12], const float n[3], float out[3]) { /* transform.h:329-334 */
    for (int i = 0; i < 3; ++i) out[i] = mi[i] * n[0] + mi[4 + i] * n[1] + mi[8 + i] * n[2];
}
void orc_transform_interaction(const float m[12], const float m_inv[12], const float in39[39], float out39[39]) {
    /* Point3fi (transform.h:133-176) */
    float x[3], err_in[3], exact = 1;
    for (int k = 0; k < 3; ++k) {
        x[k] = (in39[k] + in39[3 + k]) / 2;         /* Float(Interval) = Midpoint() */
        err_in[k] = (in39[3 + k] - in39[k]) / 2;    /* Error() = Width() / 2 */
        if (in39[3 + k] - in39[k] != 0) exact = 0;
    }
    const float g3 = orc_gamma(3);
    for (int i = 0; i < 3; ++i) {
        const float *r = m + 4 * i;
        const float p = (r[0] * x[0] + r[1] * x[1]) + (r[2] * x[2] + r[3]);
        const float a = fabsf(r[0] * x[0]) + fabsf(r[1] * x[1]) + fabsf(r[2] * x[2]) + fabsf(r[3]);
        float e;
        if (exact) e = g3 * a;
        else e = (g3 + 1) * (fabsf(r[0]) * err_in[0] + fabsf(r[1]) * err_in[1] + fabsf(r[2]) * err_in[2]) + g3 * a;
        orc_ivl iv = ivl_from_value_and_error(p, e);
        out39[i] = iv.lo;
        out39[3 + i] = iv.hi;
    }
    float t[3], n[3], ns[3];
    xf_normal(m_inv, in39 + 6, t);
    orc_normalize(t, n);                         /* ret.n = Normalize(t(si.n)) */
    xf_vec(m, in39 + 9, t);
    orc_normalize(t, out39 + 9);                 /* ret.wo */
    xf_vec(m, in39 + 12, out39 + 12);            /* dpdu */
    xf_vec(m, in39 + 15, out39 + 15);            /* dpdv */
    xf_normal(m_inv, in39 + 18, out39 + 18);     /* dndu */
    xf_normal(m_inv, in39 + 21, out39 + 21);     /* dndv */
    xf_normal(m_inv, in39 + 24, t);
    orc_normalize(t, ns);                        /* shading.n */
    xf_vec(m, in39 + 27, out39 + 27);
    xf_vec(m, in39 + 30, out39 + 30);
    xf_normal(m_inv, in39 + 33, out39 + 33);
    xf_normal(m_inv, in39 + 36, out39 + 36);
    if (orc_dot_n(ns, n) < 0.f)                  /* shading.n = FaceForward(shading.n, n) (:257) */
        for (int k = 0; k < 3; ++k) ns[k] = -ns[k];
    memcpy(out39 + 6, n, 12);
    memcpy(out39 + 24, ns, 12);
}
/* records as oracle/ref_interaction.cpp's "xf" mode: 72 floats in (4x4 m, 4x4 mInv, 39 fields, pad), 40 out */
void orc_transform_interaction_batch(const float *in72, int n, float *out40) {
    for (int i = 0; i < n; ++i) {
        const float *r = in72 + 72 * (size_t)i;
        orc_transform_interaction(r, r + 16, r + 32, out40 + 40 * (size_t)i);
        out40[40 * (size_t)i + 39] = 0;
    }
}

/* ---- film: wavefront/film.cpp:13-40 (UpdateFilm) + film.h:239-255 (RGBFilm::AddSample) --------- */
void orc_film_add_samples(double *pixels, const int32_t bounds[4], float max_component,
                          const int32_t *px, const int32_t *py, const float *rgb, int rgb_stride,
                          const float *weight, int n_per_pass, int n_passes) {
    const int x0 = bounds[0], y0 = bounds[1], x1 = bounds[2], y1 = bounds[3];
    for (int pass = 0; pass < n_passes; ++pass) {
        for (int i = 0; i < n_per_pass; ++i) {
            const int x = px[i], y = py[i];
            if (x < x0 || x >= x1 || y < y0 || y >= y1) continue; /* film.cpp:18-19 */
            const long k = (long)pass * n_per_pass + i;
            float c[3] = {rgb[k * rgb_stride], rgb[k * rgb_stride + 1], rgb[k * rgb_stride + 2]};
            const float w = weight ? weight[k] : 1.0f;
            float m = c[0]; /* film.h:245 std::max({r, g, b}) */
            if (m < c[1]) m = c[1];
            if (m < c[2]) m = c[2];
            if (m > max_component) { /* film.h:246-247 rgb *= maxComponentValue / m */
                const float s = max_component / m;
                c[0] *= s;
                c[1] *= s;
                c[2] *= s;
            }
            double *pixel = pixels + 4 * ((long)(y - y0) * (x1 - x0) + (x - x0));
            for (int ch = 0; ch < 3; ++ch) pixel[ch] += (double)(w * c[ch]); /* film.h:252-253 */
            pixel[3] += (double)w;                                            /* film.h:254 */
        }
    }
}

/* ---- KdTreeAggregate, cpu/aggregates.cpp:746-1150 ---------------------------------------------- */
/* Bounds3::IntersectP(Point3f o, Vector3f d, Float tMax, Float *hitt0, Float *hitt1),
 * util/vecmath.h:1547-1571 */
int orc_bounds_t0t1(const float bounds[6], const float o[3], const float d[3], float tmax, float t0t1[2]) {
    float t0 = 0, t1 = tmax;
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1 / d[i];
        float tNear = (bounds[i] - o[i]) * invRayDir;
        float tFar = (bounds[3 + i] - o[i]) * invRayDir;
        if (tNear > tFar) {
            float s = tNear;
            tNear = tFar;
            tFar = s;
        }
        tFar *= 1 + 2 * orc_gamma(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return 0;
    }
    t0t1[0] = t0;
    t0t1[1] = t1;
    return 1;
}

void orc_bounds_t0t1_batch(const float *bounds6, const float *o3, const float *d3, const float *tmax, int n,
                           uint8_t *hit, float *t0t1) {
    for (int i = 0; i < n; ++i) {
        t0t1[2 * i] = t0t1[2 * i + 1] = 0.0f;
        hit[i] = (uint8_t)orc_bounds_t0t1(bounds6 + 6 * i, o3 + 3 * i, d3 + 3 * i, tmax[i], t0t1 + 2 * i);
    }
}

typedef struct {
    int node;
    float tMin, tMax;
} kd_to_visit; /* KdNodeToVisit, aggregates.cpp:747-750 */

static inline float kd_split(const orc_kd_node *n) {
    float f;
    memcpy(&f, &n->split_or_index, 4);
    return f;
}

/* closest = 1: KdTreeAggregate::Intersect (:973-1067); 0: IntersectP (:1069-1150) */
static void kd_one(const orc_kd_node *nodes, const int32_t *prim_indices, const orc_prim *prims,
                   const float *verts, const float bounds[6], const orc_ray *ray, int closest, orc_hit *hit,
                   uint8_t *occ, int32_t *visited_out, int32_t *tests_out) {
    float rayTMax = ray->tmax;
    const float *o = ray->o, *d = ray->d;
    int nodesVisited = 0, tests = 0, host = 0, found = 0;
    if (closest) {
        hit->prim = -1;
        hit->t = rayTMax;
        hit->b0 = hit->b1 = hit->b2 = 0.0f;
        hit->instance = 0;
    }
    float tt[2];
    if (orc_bounds_t0t1(bounds, o, d, rayTMax, tt)) {
        float tMin = tt[0], tMax = tt[1];
        const float invDir[3] = {1 / d[0], 1 / d[1], 1 / d[2]};
        kd_to_visit toVisit[64];
        int toVisitIndex = 0;
        int node = 0;
        while (node >= 0) {
            if (closest && rayTMax < tMin) break; /* :989-991 */
            ++nodesVisited;
            const orc_kd_node *nd = &nodes[node];
            if ((nd->flags & 3) != 3) { /* interior, :993-1023 / :1110-1144 */
                const int axis = (int)(nd->flags & 3);
                const float split = kd_split(nd);
                const float tSplit = (split - o[axis]) * invDir[axis];
                const int belowFirst = (o[axis] < split) || (o[axis] == split && d[axis] <= 0);
                int firstChild, secondChild;
                if (belowFirst) {
                    firstChild = node + 1;
                    secondChild = (int)(nd->flags >> 2);
                } else {
                    firstChild = (int)(nd->flags >> 2);
                    secondChild = node + 1;
                }
                if (tSplit > tMax || tSplit <= 0)
                    node = firstChild;
                else if (tSplit < tMin)
                    node = secondChild;
                else {
                    toVisit[toVisitIndex].node = secondChild;
                    toVisit[toVisitIndex].tMin = tSplit;
                    toVisit[toVisitIndex].tMax = tMax;
                    ++toVisitIndex;
                    node = firstChild;
                    tMax = tSplit;
                }
            } else { /* leaf, :1025-1061 / :1084-1107 */
                const int nPrimitives = (int)(nd->flags >> 2);
                for (int i = 0; i < nPrimitives && !found; ++i) {
                    const int index = nPrimitives == 1 ? (int32_t)nd->split_or_index
                                                       : prim_indices[(int32_t)nd->split_or_index + i];
                    const orc_prim *p = &prims[index];
                    float r[4];
                    if (p->kind == 3) { /* host-only primitive: the record is void */
                        host = 1;
                        continue;
                    }
                    int primHit; /* GeometricPrimitive with a constant alpha: cpu/primitive.cpp:57-70, 79-81 */
                    if (p->kind >= 4 && p->kind <= 15) {
                        primHit = alpha_intersect(p, verts, o, d, rayTMax, r, &tests, &host);
                    } else {
                        ++tests;
                        primHit = prim_test(p, verts, o, d, rayTMax, r);
                    }
                    if (primHit) {
                        if (closest) {
                            hit->prim = p->id;
                            hit->b0 = r[0];
                            hit->b1 = r[1];
                            hit->b2 = r[2];
                            hit->t = r[3];
                            rayTMax = r[3];
                        } else {
                            found = 1; /* :1091-1094, :1101-1104: return true */
                        }
                    }
                }
                if (found) break;
                if (toVisitIndex > 0) {
                    --toVisitIndex;
                    node = toVisit[toVisitIndex].node;
                    tMin = toVisit[toVisitIndex].tMin;
                    tMax = toVisit[toVisitIndex].tMax;
                } else
                    break;
            }
        }
    }
    if (closest) {
        hit->nodes_visited = nodesVisited;
        hit->prim_tests = tests;
        if (host) hit->instance = -1;
    } else {
        *occ = found ? 1 : (host ? 2 : 0);
        if (visited_out) *visited_out = nodesVisited;
        if (tests_out) *tests_out = tests;
    }
}

typedef struct {
    const orc_kd_node *nodes;
    const int32_t *prim_indices;
    const orc_prim *prims;
    const float *verts;
    const float *bounds;
    const orc_ray *rays;
    orc_hit *hits;
    uint8_t *occ;
    int32_t *visited, *tests;
    int64_t begin, end;
} kd_job;

static void *kd_worker(void *arg) {
    kd_job *j = (kd_job *)arg;
    for (int64_t i = j->begin; i < j->end; ++i) {
        if (j->hits)
            kd_one(j->nodes, j->prim_indices, j->prims, j->verts, j->bounds, &j->rays[i], 1, &j->hits[i], NULL,
                   NULL, NULL);
        else
            kd_one(j->nodes, j->prim_indices, j->prims, j->verts, j->bounds, &j->rays[i], 0, NULL, &j->occ[i],
                   j->visited ? &j->visited[i] : NULL, j->tests ? &j->tests[i] : NULL);
    }
    return NULL;
}

static void kd_run(kd_job base, int64_t n, int nthreads) {
    if (nthreads <= 1 || n < 2 * (int64_t)nthreads) {
        base.begin = 0;
        base.end = n;
        kd_worker(&base);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    kd_job *jobs = (kd_job *)malloc(sizeof(kd_job) * (size_t)nthreads);
    for (int k = 0; k < nthreads; ++k) {
        jobs[k] = base;
        jobs[k].begin = n * k / nthreads;
        jobs[k].end = n * (k + 1) / nthreads;
        pthread_create(&th[k], NULL, kd_worker, &jobs[k]);
    }
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    free(jobs);
    free(th);
}

void orc_kd_intersect_closest(const orc_kd_node *nodes, const int32_t *prim_indices, const orc_prim *prims,
                              const float *verts, const float bounds[6], const orc_ray *rays, int64_t n,
                              orc_hit *hits, int nthreads) {
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    kd_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prim_indices = prim_indices;
    j.prims = prims;
    j.verts = verts;
    j.bounds = bounds;
    j.rays = rays;
    j.hits = hits;
    kd_run(j, n, nthreads);
}

void orc_kd_intersect_any(const orc_kd_node *nodes, const int32_t *prim_indices, const orc_prim *prims,
                          const float *verts, const float bounds[6], const orc_ray *rays, int64_t n,
                          uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests, int nthreads) {
    g_prim_alpha_base = prims; /* alpha-tested patches index orc_set_prim_alpha's array by position */
    kd_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes;
    j.prim_indices = prim_indices;
    j.prims = prims;
    j.verts = verts;
    j.bounds = bounds;
    j.rays = rays;
    j.occ = occluded;
    j.visited = nodes_visited;
    j.tests = prim_tests;
    kd_run(j, n, nthreads);
}

/* ---- util/hash.h ------------------------------------------------------------------------------ */
static uint64_t orc_murmur64a(const unsigned char *key, size_t len, uint64_t seed) { /* hash.h:19-64 */
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = seed ^ (len * m);
    const unsigned char *end = key + 8 * (len / 8);
    while (key != end) {
        uint64_t k;
        memcpy(&k, key, sizeof(uint64_t));
        key += 8;
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    switch (len & 7) {
    case 7: h ^= (uint64_t)key[6] << 48; /* fall through */
    case 6: h ^= (uint64_t)key[5] << 40; /* fall through */
    case 5: h ^= (uint64_t)key[4] << 32; /* fall through */
    case 4: h ^= (uint64_t)key[3] << 24; /* fall through */
    case 3: h ^= (uint64_t)key[2] << 16; /* fall through */
    case 2: h ^= (uint64_t)key[1] << 8;  /* fall through */
    case 1: h ^= (uint64_t)key[0]; h *= m;
    };
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

uint64_t orc_hash_6f(const float a[3], const float b[3]) { /* Hash(args...), hash.h:111-118 */
    uint64_t buf[3];
    memcpy((char *)buf, a, 12);
    memcpy((char *)buf + 12, b, 12);
    return orc_murmur64a((const unsigned char *)buf, 24, 0);
}

float orc_hash_float_6f(const float a[3], const float b[3]) { /* hash.h:120-123 */
    return (float)(uint32_t)orc_hash_6f(a, b) * 0x1p-32f;
}

static uint64_t orc_mix_bits(uint64_t v) { /* hash.h:70-77 */
    v ^= (v >> 31);
    v *= 0x7fb5d329728ea185ull;
    v ^= (v >> 27);
    v *= 0x81dadef4bc2dd44dull;
    v ^= (v >> 33);
    return v;
}

/* ---- util/rng.h: PCG32 -------------------------------------------------------------------------- */
typedef struct {
    uint64_t state, inc;
} orc_rng;
static uint32_t orc_rng_u32(orc_rng *g) { /* rng.h:92-99 */
    uint64_t oldstate = g->state;
    g->state = oldstate * 0x5851f42d4c957f2dull + g->inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
static void orc_rng_set_sequence(orc_rng *g, uint64_t sequenceIndex) { /* rng.h:49-51, 130-136 */
    const uint64_t seed = orc_mix_bits(sequenceIndex);
    g->state = 0u;
    g->inc = (sequenceIndex << 1u) | 1u;
    orc_rng_u32(g);
    g->state += seed;
    orc_rng_u32(g);
}
static float orc_rng_float(orc_rng *g) { /* rng.h:138-141 */
    const float v = (float)orc_rng_u32(g) * 0x1p-32f;
    return v < 0x1.fffffep-1f ? v : 0x1.fffffep-1f;
}

int orc_wrs_unit_weights(uint64_t seed, int n_adds, float *sample_probability, float *weight_sum) {
    orc_rng g; /* WeightedReservoirSampler(uint64_t rngSeed) : rng(rngSeed), sampling.h:529 */
    orc_rng_set_sequence(&g, seed);
    float weightSum = 0, reservoirWeight = 0;
    int reservoir = -1;
    for (int k = 0; k < n_adds; ++k) { /* Add(sample, weight), sampling.h:535-546 */
        const float weight = 1.f;
        weightSum += weight;
        const float p = weight / weightSum;
        if (orc_rng_float(&g) < p) {
            reservoir = k;
            reservoirWeight = weight;
        }
    }
    if (sample_probability) *sample_probability = weightSum > 0 ? reservoirWeight / weightSum : 0.f;
    if (weight_sum) *weight_sum = weightSum;
    return weightSum > 0 ? reservoir : -1;
}

/* ---- ray.h:75-101 ------------------------------------------------------------------------------- */
void orc_offset_ray_origin(const float lo[3], const float hi[3], const float n[3], const float w[3],
                           float po[3]) {
    /* pi.Error() = Width() / 2 (vecmath.h Point3fi::Error, math.h Interval::Width = high - low) */
    const float ex = (hi[0] - lo[0]) / 2, ey = (hi[1] - lo[1]) / 2, ez = (hi[2] - lo[2]) / 2;
    /* Dot(Normal3, Vector3) = FMA(n.x, v.x, SumOfProducts(n.y, v.y, n.z, v.z)), vecmath.h:1056-1068 */
    const float an[3] = {fabsf(n[0]), fabsf(n[1]), fabsf(n[2])}, err[3] = {ex, ey, ez};
    const float d = orc_dot_n(an, err);
    float offset[3] = {d * n[0], d * n[1], d * n[2]};
    if (orc_dot_n(n, w) < 0) {
        offset[0] = -offset[0];
        offset[1] = -offset[1];
        offset[2] = -offset[2];
    }
    for (int i = 0; i < 3; ++i) {
        po[i] = (lo[i] + hi[i]) / 2 + offset[i]; /* Point3f(pi): Interval::Midpoint */
        if (offset[i] > 0)
            po[i] = next_up(po[i]);
        else if (offset[i] < 0)
            po[i] = next_down(po[i]);
    }
}

void orc_spawn_ray_to(const float lo[3], const float hi[3], const float n[3], const float p_to[3],
                      float out_o[3], float out_d[3]) {
    for (int i = 0; i < 3; ++i) out_d[i] = p_to[i] - (lo[i] + hi[i]) / 2; /* ray.h:99 */
    orc_offset_ray_origin(lo, hi, n, out_d, out_o);
}

void orc_hash_batch(const float *in6, int n, uint32_t *lo, uint32_t *hi, float *hash_float) {
    for (int i = 0; i < n; ++i) {
        const uint64_t h = orc_hash_6f(in6 + 6 * i, in6 + 6 * i + 3);
        lo[i] = (uint32_t)h;
        hi[i] = (uint32_t)(h >> 32);
        hash_float[i] = orc_hash_float_6f(in6 + 6 * i, in6 + 6 * i + 3);
    }
}

void orc_offset_batch(const float *in12, int n, float *out9) {
    for (int i = 0; i < n; ++i) {
        const float *r = in12 + 12 * i;
        orc_offset_ray_origin(r, r + 3, r + 6, r + 9, out9 + 9 * i);
        orc_spawn_ray_to(r, r + 3, r + 6, r + 9, out9 + 9 * i + 3, out9 + 9 * i + 6);
    }
}

void orc_wrs_batch(const float *in7, int n, int32_t *selected, float *out2) {
    for (int i = 0; i < n; ++i) {
        const float *r = in7 + 7 * i;
        selected[i] = orc_wrs_unit_weights(orc_hash_6f(r, r + 3), (int)r[6], &out2[2 * i], &out2[2 * i + 1]);
    }
}
