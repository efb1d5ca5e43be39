// trace_math.h — device arithmetic shared by the traversal kernels (bvh_trace.hip, kd_trace.hip):
// the reference's leaf tests and helpers restated operation for operation.
//   slab test     Bounds3::IntersectP       util/vecmath.h:1573-1608
//   triangle      IntersectTriangle         shapes.cpp:172-273
//   patch         IntersectBilinearPatch    shapes.h:1279-1347 (+ util/math.h:614-637, 1420-1426)
//   ray transform Transform::ApplyInverse   util/transform.h:416-429
// Compiled with -ffp-contract=off; __builtin_fmaf exactly where the reference calls FMA().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnbvh {

#define DEV static __device__ __forceinline__

// util/float.h:43,195-197 — evaluated in float exactly as the reference's constexpr
DEV constexpr float gamma_f(int n) {
    return ((float)n * 0x1p-24f) / (1.0f - (float)n * 0x1p-24f);
}

// util/math.h:569-575
DEV float dop(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = __builtin_fmaf(a, b, -cd);
    float err = __builtin_fmaf(-c, d, cd);
    return diff + err;
}

DEV float fmax3(float a, float b, float c) {  // v_max3_f32: a NaN operand is ignored
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}
DEV float max3(float a, float b, float c) {  // std::max({a,b,c})
    float m = a;
    if (m < b) m = b;
    if (m < c) m = c;
    return m;
}

struct V3 {
    float x, y, z;
};
DEV V3 cross(V3 v, V3 w) {  // util/vecmath.h:999-1004
    return {dop(v.y, w.z, v.z, w.y), dop(v.z, w.x, v.x, w.z), dop(v.x, w.y, v.y, w.x)};
}
DEV float dot(V3 v, V3 w) { return v.x * w.x + v.y * w.y + v.z * w.z; }
DEV float len2(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
DEV V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV float maxabs(V3 v) { return max3(__builtin_fabsf(v.x), __builtin_fabsf(v.y), __builtin_fabsf(v.z)); }
DEV float sel3(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

struct RayState {
    V3 o, inv;  // dirIsNeg[k] (aggregates.cpp:535) is recomputed as inv.k < 0 where needed
                // (the direction itself is cold state: only the patch test reads it, from LDS)
    // The part of IntersectTriangle that depends on the ray only (shapes.cpp:186-201): the
    // permutation (kz = largest |d| component) and the shear Sx, Sy, Sz.  The reference
    // recomputes them for every triangle; computed once per ray they are the same floats.
    float sx, sy, sz;
    int kz;  // kz << 8 | dirIsNeg bits (bit k = inv.k < 0, aggregates.cpp:535): see ray_shear
};

DEV void ray_shear(RayState &r, V3 d) {
    float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);  // vecmath.h:453-455
    int kx = kz + 1;
    if (kx == 3) kx = 0;
    int ky = kx + 1;
    if (ky == 3) ky = 0;
    float dx = kx == 0 ? d.x : (kx == 1 ? d.y : d.z);
    float dy = ky == 0 ? d.x : (ky == 1 ? d.y : d.z);
    float dz = kz == 0 ? d.x : (kz == 1 ? d.y : d.z);
    r.sx = -dx / dz;
    r.sy = -dy / dz;
    r.sz = 1.0f / dz;
    // r.inv is set by now: the three dirIsNeg flags ride in the low bits, for the interior step's
    // `dirIsNeg[node->axis]` (one v_bfe_u32)
    r.kz = (kz << 8) | (r.inv.x < 0.0f ? 1 : 0) | (r.inv.y < 0.0f ? 2 : 0) | (r.inv.z < 0.0f ? 4 : 0);
}

// Wave-level select masks of the ray-invariant predicates a step would otherwise recompute per lane and per
// step: bit l of a mask = the predicate for lane l's ray (a ballot, held in an SGPR pair).  Valid until a
// lane's ray changes; the lean traversal kernels rebuild them after every refill trip.
struct RayMasks {
    unsigned long long negX, negY, negZ;  // inv.k < 0  (dirIsNeg, aggregates.cpp:535)
    unsigned long long k0, k1, k2;        // kz == 0 / 1 / 2  (shapes.cpp:186)
};
DEV RayMasks ray_masks(const RayState &r) {
    const int kz = r.kz >> 8;
    return {__ballot(r.inv.x < 0.0f), __ballot(r.inv.y < 0.0f), __ballot(r.inv.z < 0.0f),
            __ballot(kz == 0), __ballot(kz == 1), __ballot(kz == 2)};
}
DEV float sel_m(unsigned long long m, float a, float b) {  // lane's bit of m set ? a : b — one v_cndmask, no compare
    float out;
    asm("v_cndmask_b32_e64 %0, %2, %1, %3" : "=v"(out) : "v"(a), "v"(b), "s"(m));
    return out;
}

// Slab test of util/vecmath.h:1573-1608 split in two: everything that does not involve
// the ray's tMax is evaluated here (`early` = none of the reference's early-outs fired and
// box tMax > 0), and the entry distance is returned so that the remaining conjunct
// `tMin < raytMax` can be evaluated now (near child) or when the node is popped (far child).
// Branch-free: the same comparisons on the same values as the reference, combined without
// short-circuiting (values computed past a fired early-out are simply not used).
DEV bool slab_partial(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                      const RayState &r, float &tEntry) {
    constexpr float widen = 1.0f + 2.0f * gamma_f(3);
    float tMin = (((r.inv.x < 0.0f) ? mxx : mnx) - r.o.x) * r.inv.x;
    float tMax = (((r.inv.x < 0.0f) ? mnx : mxx) - r.o.x) * r.inv.x;
    float tyMin = (((r.inv.y < 0.0f) ? mxy : mny) - r.o.y) * r.inv.y;
    float tyMax = (((r.inv.y < 0.0f) ? mny : mxy) - r.o.y) * r.inv.y;
    tMax *= widen;
    tyMax *= widen;
    const bool out1 = (tMin > tyMax) | (tyMin > tMax);
    tMin = (tyMin > tMin) ? tyMin : tMin;
    tMax = (tyMax < tMax) ? tyMax : tMax;
    float tzMin = (((r.inv.z < 0.0f) ? mxz : mnz) - r.o.z) * r.inv.z;
    float tzMax = (((r.inv.z < 0.0f) ? mnz : mxz) - r.o.z) * r.inv.z;
    tzMax *= widen;
    const bool out2 = (tMin > tzMax) | (tzMin > tMax);
    tMin = (tzMin > tMin) ? tzMin : tMin;
    tMax = (tzMax < tMax) ? tzMax : tMax;
    tEntry = tMin;
    return !(out1 | out2) & (tMax > 0.0f);
}

// The same test reduced to ONE float per box: the entry distance if every test of the reference
// other than `tMin < raytMax` passes, +inf otherwise — so that `key < raytMax` IS the reference's
// verdict, now for the near child and later (against the then-current tMax) for a far child.
//
// With a_k / b_k the per-axis entry / widened exit distances (k = x, y, z) the reference rejects iff
// a_x > b_y, a_y > b_x (first early-out), max'(a_x, a_y) > b_z or a_z > min'(b_x, b_y) (second), i.e.
// iff a_i > b_j for some i != j, where max' / min' keep their first operand when the other is NaN;
// it accepts iff additionally max'(a) < raytMax and min'(b) > 0.  Three facts make that equal to
// `x ordered  &&  max'(a) <= min'(b)  &&  min'(b) > 0  &&  max'(a) < raytMax`:
//  * a NaN a_x or b_x always ends in a miss (every comparison it enters is false, and it survives
//    into the final tMin / tMax), while a NaN in y or z is ignored by both forms: v_max_f32 /
//    v_min_f32 return the other operand, as the reference's `if (ty > t) t = ty` does;
//  * the only pairs the second form adds are the diagonal ones, a_k > b_k.  For a box with
//    min <= max (checked when the scene is created) rounding is monotone, so a_k <= the unwidened
//    exit distance; widening by 1 + 2 gamma(3) only moves b_k below a_k when b_k < 0, and then
//    min'(b) > 0 fails in both forms;
//  * max' and v_max_f32 may differ in the sign of a zero, which no comparison sees.
// tools/slab_equivalence (tests/test_slab_equivalence.py) checks the two forms against each other on
// adversarial boxes and rays (zero / infinite / NaN components, origins on slab planes).
DEV float slab_entry_key(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const RayState &r) {
    constexpr float widen = 1.0f + 2.0f * gamma_f(3);
    const float ax = (((r.inv.x < 0.0f) ? mxx : mnx) - r.o.x) * r.inv.x;
    const float bx = (((r.inv.x < 0.0f) ? mnx : mxx) - r.o.x) * r.inv.x;
    const float ay = (((r.inv.y < 0.0f) ? mxy : mny) - r.o.y) * r.inv.y;
    const float by = (((r.inv.y < 0.0f) ? mny : mxy) - r.o.y) * r.inv.y;
    const float az = (((r.inv.z < 0.0f) ? mxz : mnz) - r.o.z) * r.inv.z;
    const float bz = (((r.inv.z < 0.0f) ? mnz : mxz) - r.o.z) * r.inv.z;
    const float a = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    // the reference widens each exit distance (t *= 1 + 2 gamma(3)) before it takes their minimum; rounded
    // multiplication by a positive constant is monotone and keeps a NaN a NaN, so the minimum of the widened
    // values IS the widened minimum: one multiplication instead of three
    const float b = __builtin_fminf(__builtin_fminf(bx, by), bz) * widen;
    const bool ok = !__builtin_isunordered(ax, bx) & (a <= b) & (b > 0.0f);
    return ok ? a : __builtin_inff();
}

// slab_entry_key with the three sign tests taken from the wave's masks
DEV float slab_entry_key(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const RayState &r,
                         const RayMasks &m) {
    constexpr float widen = 1.0f + 2.0f * gamma_f(3);
    const float ax = (sel_m(m.negX, mxx, mnx) - r.o.x) * r.inv.x;
    const float bx = (sel_m(m.negX, mnx, mxx) - r.o.x) * r.inv.x;
    const float ay = (sel_m(m.negY, mxy, mny) - r.o.y) * r.inv.y;
    const float by = (sel_m(m.negY, mny, mxy) - r.o.y) * r.inv.y;
    const float az = (sel_m(m.negZ, mxz, mnz) - r.o.z) * r.inv.z;
    const float bz = (sel_m(m.negZ, mnz, mxz) - r.o.z) * r.inv.z;
    const float a = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    const float b = __builtin_fminf(__builtin_fminf(bx, by), bz) * widen;
    const bool ok = !__builtin_isunordered(ax, bx) & (a <= b) & (b > 0.0f);
    return ok ? a : __builtin_inff();
}

// shapes.cpp:172-273.  `degenerate` is the reference's first test
// (LengthSquared(Cross(p2 - p0, p1 - p0)) == 0, :176-177): it depends on the triangle only and
// is evaluated once, with the same float32 operations, when the scene is baked (kPrimDegenerate).
DEV bool triangle_test(const RayState &r, float tMax, bool degenerate, V3 p0, V3 p1, V3 p2,
                       float &b0, float &b1, float &b2, float &tHit, const RayMasks *m = nullptr) {
    if (degenerate) return false;
    V3 a = sub(p0, r.o), b = sub(p1, r.o), c = sub(p2, r.o);
    float p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y, p2z;
    if (m) {
        // kx = kz + 1 mod 3, ky = kx + 1 mod 3: component kx is x iff kz == 2, y iff kz == 0, else z; component
        // ky is x iff kz == 1, y iff kz == 2, else z — the same selections as sel3, from the wave's masks
        auto px = [&](V3 v) { return sel_m(m->k2, v.x, sel_m(m->k0, v.y, v.z)); };
        auto py = [&](V3 v) { return sel_m(m->k1, v.x, sel_m(m->k2, v.y, v.z)); };
        auto pz = [&](V3 v) { return sel_m(m->k0, v.x, sel_m(m->k1, v.y, v.z)); };
        p0x = px(a), p0y = py(a), p0z = pz(a);
        p1x = px(b), p1y = py(b), p1z = pz(b);
        p2x = px(c), p2y = py(c), p2z = pz(c);
    } else {
        const int kz = r.kz >> 8;
        int kx = kz + 1;
        if (kx == 3) kx = 0;
        int ky = kx + 1;
        if (ky == 3) ky = 0;
        p0x = sel3(a, kx), p0y = sel3(a, ky), p0z = sel3(a, kz);
        p1x = sel3(b, kx), p1y = sel3(b, ky), p1z = sel3(b, kz);
        p2x = sel3(c, kx), p2y = sel3(c, ky), p2z = sel3(c, kz);
    }
    const float sx = r.sx, sy = r.sy, sz = r.sz;
    p0x += sx * p0z;
    p0y += sy * p0z;
    p1x += sx * p1z;
    p1y += sy * p1z;
    p2x += sx * p2z;
    p2y += sy * p2z;
    float e0 = dop(p1x, p2y, p1y, p2x);
    float e1 = dop(p2x, p0y, p2y, p0x);
    float e2 = dop(p0x, p1y, p0y, p1x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // :215-225, fp64 on device
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
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0z *= sz;
    p1z *= sz;
    p2z *= sz;
    float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1.0f / det;
    float t = tScaled * invDet;
    // std::max({a, b, c}) (vecmath.h:448-451) is NaN iff its FIRST element is (every later
    // `largest < x` is then false) and ignores a NaN second or third element; v_max3_f32 ignores a NaN
    // anywhere.  The four maxima reach nothing but deltaT, which a NaN maximum turns into NaN, and
    // `t <= NaN` is false: one flag over the four first elements restores the difference.
    const bool firstIsNan = __builtin_isunordered(p0z, p0x) | __builtin_isunordered(p0y, e0);
    float maxZt = fmax3(__builtin_fabsf(p0z), __builtin_fabsf(p1z), __builtin_fabsf(p2z));
    float deltaZ = gamma_f(3) * maxZt;
    float maxXt = fmax3(__builtin_fabsf(p0x), __builtin_fabsf(p1x), __builtin_fabsf(p2x));
    float maxYt = fmax3(__builtin_fabsf(p0y), __builtin_fabsf(p1y), __builtin_fabsf(p2y));
    float deltaX = gamma_f(5) * (maxXt + maxZt);
    float deltaY = gamma_f(5) * (maxYt + maxZt);
    float deltaE = 2.0f * (gamma_f(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = fmax3(__builtin_fabsf(e0), __builtin_fabsf(e1), __builtin_fabsf(e2));
    float deltaT = 3.0f * (gamma_f(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) *
                   __builtin_fabsf(invDet);
    if (!firstIsNan & (t <= deltaT)) return false;
    b0 = e0 * invDet;
    b1 = e1 * invDet;
    b2 = e2 * invDet;
    tHit = t;
    return true;
}

// util/math.h:614-637
DEV bool quadratic(float a, float b, float c, float &t0, float &t1) {
    if (a == 0) {
        if (b == 0) return false;
        t0 = t1 = -c / b;
        return true;
    }
    float discrim = dop(b, b, 4.0f * a, c);
    if (discrim < 0) return false;
    float root = __builtin_sqrtf(discrim);
    float q = -0.5f * (b + __builtin_copysignf(root, b));
    t0 = q / a;
    t1 = c / q;
    if (t0 > t1) {
        float s = t0;
        t0 = t1;
        t1 = s;
    }
    return true;
}

// util/math.h:1420-1426 with rows (r0, r1, r2)
DEV float det3(V3 r0, V3 r1, V3 r2) {
    float minor12 = dop(r1.y, r2.z, r1.z, r2.y);
    float minor02 = dop(r1.x, r2.z, r1.z, r2.x);
    float minor01 = dop(r1.x, r2.y, r1.y, r2.x);
    return __builtin_fmaf(r0.z, minor01, dop(r0.x, minor12, r0.y, minor02));
}

DEV V3 lerp3(float t, V3 a, V3 b) {  // (1 - t) * a + t * b, util/vecmath.h:410-412
    float omt = 1.0f - t;
    return {omt * a.x + t * b.x, omt * a.y + t * b.y, omt * a.z + t * b.z};
}

DEV void patch_root(float u, const RayState &r, V3 rd, V3 p00, V3 p10, V3 p01, V3 p11, float &vnum,
                    float &tnum, float &p2) {
    V3 uo = lerp3(u, p00, p10);
    V3 ud = sub(lerp3(u, p01, p11), uo);
    V3 deltao = sub(uo, r.o);
    V3 perp = cross(rd, ud);
    p2 = len2(perp);
    vnum = det3({deltao.x, rd.x, perp.x}, {deltao.y, rd.y, perp.y}, {deltao.z, rd.z, perp.z});
    tnum = det3({deltao.x, ud.x, perp.x}, {deltao.y, ud.y, perp.y}, {deltao.z, ud.z, perp.z});
}

// shapes.h:1279-1347
DEV bool patch_test(const RayState &r, V3 rd, float tMax, V3 p00, V3 p10, V3 p01, V3 p11, float &uOut,
                    float &vOut, float &tOut) {
    float a = dot(cross(sub(p10, p00), sub(p01, p11)), rd);
    float c = dot(cross(sub(p00, r.o), rd), sub(p01, p00));
    float b = dot(cross(sub(p10, r.o), rd), sub(p11, p10)) - (a + c);
    float u1, u2;
    if (!quadratic(a, b, c, u1, u2)) return false;
    float eps = gamma_f(10) * (maxabs(r.o) + maxabs(rd) + maxabs(p00) + maxabs(p10) +
                               maxabs(p01) + maxabs(p11));
    float t = tMax, u = 0.0f, v = 0.0f;
    if (0 <= u1 && u1 <= 1) {
        float v1, t1, p2;
        patch_root(u1, r, rd, p00, p10, p01, p11, v1, t1, p2);
        if (t1 > p2 * eps && 0 <= v1 && v1 <= p2) {
            u = u1;
            v = v1 / p2;
            t = t1 / p2;
        }
    }
    if (0 <= u2 && u2 <= 1 && u2 != u1) {
        float v2, t2, p2;
        patch_root(u2, r, rd, p00, p10, p01, p11, v2, t2, p2);
        t2 /= p2;
        if (0 <= v2 && v2 <= p2 && t > t2 && t2 > eps) {
            t = t2;
            u = u2;
            v = v2 / p2;
        }
    }
    if (t >= tMax) return false;
    uOut = u;
    vOut = v;
    tOut = t;
    return true;
}

// ---- TransformedPrimitive's ray transform (two-level scenes) ------------------------------
// Transform::ApplyInverse(const Ray&, Float *tMax), util/transform.h:416-429, on top of
// Transform::ApplyInverse(const Point3fi&), util/transform.cpp:263-303 (exact-input branch), with
// the CPU forms of the directed-rounding helpers (util/float.h:163-260: NextFloatUp/Down of the
// round-to-nearest result) and Interval::{FromValueAndError, +=, /, Midpoint, Width}
// (util/math.h:829-853, 874-876, 1028-1036).
DEV float next_up(float v) {
    if (__builtin_isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    unsigned ui = __float_as_uint(v);
    if (v >= 0) ++ui;
    else --ui;
    return __uint_as_float(ui);
}
DEV float next_down(float v) {
    if (__builtin_isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    unsigned ui = __float_as_uint(v);
    if (v > 0) --ui;
    else ++ui;
    return __uint_as_float(ui);
}
struct Ivl {
    float lo, hi;
};
DEV Ivl ivl(float a, float b) { return {(b < a) ? b : a, (a < b) ? b : a}; }
DEV Ivl ivl_from_value_and_error(float v, float err) {
    if (err == 0) return {v, v};
    return {next_down(v + (-err)), next_up(v + err)};
}
DEV Ivl ivl_add_f(Ivl a, float f) { return ivl(next_down(a.lo + f), next_up(a.hi + f)); }
DEV Ivl ivl_div_f(Ivl i, float f) {
    if (f == 0) return ivl(-__builtin_inff(), __builtin_inff());
    if (f > 0) return ivl(next_down(i.lo / f), next_up(i.hi / f));
    return ivl(next_down(i.hi / f), next_up(i.lo / f));
}
// rows r0, r1, r2 of the 3x4 inverse matrix
DEV void apply_inverse_ray(float4 r0, float4 r1, float4 r2, V3 o, V3 d, float &tMax, V3 &oOut,
                           V3 &dOut) {
    constexpr float g3 = gamma_f(3);
    const float px = (r0.x * o.x + r0.y * o.y) + (r0.z * o.z + r0.w);
    const float py = (r1.x * o.x + r1.y * o.y) + (r1.z * o.z + r1.w);
    const float pz = (r2.x * o.x + r2.y * o.y) + (r2.z * o.z + r2.w);
    const float ex0 = g3 * (__builtin_fabsf(r0.x * o.x) + __builtin_fabsf(r0.y * o.y) + __builtin_fabsf(r0.z * o.z));
    const float ey0 = g3 * (__builtin_fabsf(r1.x * o.x) + __builtin_fabsf(r1.y * o.y) + __builtin_fabsf(r1.z * o.z));
    const float ez0 = g3 * (__builtin_fabsf(r2.x * o.x) + __builtin_fabsf(r2.y * o.y) + __builtin_fabsf(r2.z * o.z));
    const float wp = (0.f * o.x + 0.f * o.y) + (0.f * o.z + 1.f);
    Ivl xp = ivl_from_value_and_error(px, ex0), yp = ivl_from_value_and_error(py, ey0),
        zp = ivl_from_value_and_error(pz, ez0);
    if (!(wp == 1)) {
        xp = ivl_div_f(xp, wp);
        yp = ivl_div_f(yp, wp);
        zp = ivl_div_f(zp, wp);
    }
    const float dx = r0.x * d.x + r0.y * d.y + r0.z * d.z;
    const float dy = r1.x * d.x + r1.y * d.y + r1.z * d.z;
    const float dz = r2.x * d.x + r2.y * d.y + r2.z * d.z;
    const float len2 = dx * dx + dy * dy + dz * dz;
    if (len2 > 0) {
        const float ex = (xp.hi - xp.lo) / 2, ey = (yp.hi - yp.lo) / 2, ez = (zp.hi - zp.lo) / 2;
        const float dt = (__builtin_fabsf(dx) * ex + __builtin_fabsf(dy) * ey + __builtin_fabsf(dz) * ez) / len2;
        xp = ivl_add_f(xp, dx * dt);
        yp = ivl_add_f(yp, dy * dt);
        zp = ivl_add_f(zp, dz * dt);
        tMax -= dt;
    }
    oOut = {(xp.lo + xp.hi) / 2, (yp.lo + yp.hi) / 2, (zp.lo + zp.hi) / 2};
    dOut = {dx, dy, dz};
}

}  // namespace nnbvh
