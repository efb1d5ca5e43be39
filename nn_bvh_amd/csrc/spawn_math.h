// spawn_math.h — device restatements of the small reference functions the batched callers need
// around a trace (all integer / fp32, operation for operation; pinned through the oracle to the
// compiled reference by tests/test_aux_oracle.py):
//   Hash / HashFloat / MixBits          util/hash.h:19-128
//   RNG (PCG32) SetSequence / Uniform   util/rng.h:49-51, 92-99, 130-141
//   OffsetRayOrigin / SpawnRayTo        ray.h:75-101 (Dot with a Normal3 = FMA + SumOfProducts,
//                                       util/vecmath.h:1056-1068, util/math.h:577-583)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trace_math.h"

namespace nnbvh {

DEV uint64_t murmur64a_24(const uint32_t w[6]) {  // MurmurHash64A(key, 24, 0): three 8-byte blocks, no tail
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = 0ull ^ (24ull * m);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        uint64_t k = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);  // little-endian memcpy
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

// Hash(Point3f a, Vector3f / Point3f b), hash.h:111-118
DEV uint64_t hash_6f(V3 a, V3 b) {
    const uint32_t w[6] = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z),
                           __float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z)};
    return murmur64a_24(w);
}
DEV float hash_float_6f(V3 a, V3 b) { return (float)(uint32_t)hash_6f(a, b) * 0x1p-32f; }  // hash.h:120-123

DEV uint64_t mix_bits(uint64_t v) {  // hash.h:70-77
    v ^= (v >> 31);
    v *= 0x7fb5d329728ea185ull;
    v ^= (v >> 27);
    v *= 0x81dadef4bc2dd44dull;
    v ^= (v >> 33);
    return v;
}

struct Pcg32 {
    uint64_t state, inc;
};
DEV uint32_t pcg32_u32(Pcg32 &g) {  // rng.h:92-99
    const uint64_t oldstate = g.state;
    g.state = oldstate * 0x5851f42d4c957f2dull + g.inc;
    const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    const uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
DEV void pcg32_set_sequence(Pcg32 &g, uint64_t sequenceIndex) {  // rng.h:49-51, 130-136
    const uint64_t seed = mix_bits(sequenceIndex);
    g.state = 0u;
    g.inc = (sequenceIndex << 1u) | 1u;
    pcg32_u32(g);
    g.state += seed;
    pcg32_u32(g);
}
DEV float pcg32_float(Pcg32 &g) {  // rng.h:138-141: min(OneMinusEpsilon, u32 * 2^-32)
    const float v = (float)pcg32_u32(g) * 0x1p-32f;
    return v < 0x1.fffffep-1f ? v : 0x1.fffffep-1f;
}

DEV float sop(float a, float b, float c, float d) {  // SumOfProducts, math.h:577-583
    const float cd = c * d;
    const float s = __builtin_fmaf(a, b, cd);
    const float err = __builtin_fmaf(c, d, -cd);
    return s + err;
}
DEV float dot_n(V3 n, V3 v) { return __builtin_fmaf(n.x, v.x, sop(n.y, v.y, n.z, v.z)); }  // vecmath.h:1056-1068

// OffsetRayOrigin(Point3fi pi, Normal3f n, Vector3f w), ray.h:75-92
DEV V3 offset_ray_origin(V3 lo, V3 hi, V3 n, V3 w) {
    const V3 err = {(hi.x - lo.x) / 2, (hi.y - lo.y) / 2, (hi.z - lo.z) / 2};  // pi.Error()
    const V3 an = {__builtin_fabsf(n.x), __builtin_fabsf(n.y), __builtin_fabsf(n.z)};
    const float d = dot_n(an, err);
    V3 offset = {d * n.x, d * n.y, d * n.z};
    if (dot_n(n, w) < 0) offset = {-offset.x, -offset.y, -offset.z};
    V3 po = {(lo.x + hi.x) / 2 + offset.x, (lo.y + hi.y) / 2 + offset.y, (lo.z + hi.z) / 2 + offset.z};
    if (offset.x > 0) po.x = next_up(po.x);
    else if (offset.x < 0) po.x = next_down(po.x);
    if (offset.y > 0) po.y = next_up(po.y);
    else if (offset.y < 0) po.y = next_down(po.y);
    if (offset.z > 0) po.z = next_up(po.z);
    else if (offset.z < 0) po.z = next_down(po.z);
    return po;
}

// SpawnRayTo(Point3fi pFrom, Normal3f n, Float time, Point3f pTo), ray.h:98-101
DEV void spawn_ray_to(V3 lo, V3 hi, V3 n, V3 pTo, V3 &o, V3 &d) {
    d = {pTo.x - (lo.x + hi.x) / 2, pTo.y - (lo.y + hi.y) / 2, pTo.z - (lo.z + hi.z) / 2};
    o = offset_ray_origin(lo, hi, n, d);
}

// Origin of the ray GeometricPrimitive::Intersect traces on after an alpha-rejected triangle hit
// (cpu/primitive.cpp:64-66: rNext = si->intr.SpawnRay(r.d)): Triangle::InteractionFromIntersection
// (shapes.h:884-1010) gives
//   pHit = b0 p0 + b1 p1 + b2 p2, pError = gamma(7) (|b0 p0| + |b1 p1| + |b2 p2|), pi = Point3fi(pHit, pError)
//   n = Normalize(Cross(p0 - p2, p1 - p2)), negated under reverseOrientation ^ transformSwapsHandedness
// and Interaction::SpawnRay(d) = OffsetRayOrigin(pi, n, d) (interaction.h:98-100, ray.h:75-92).
// smooth = the mesh has per-vertex shading normals n0 n1 n2: the interaction's normal is then
//   FaceForward(n, ns), ns = Normalize(b0 n0 + b1 n1 + b2 n2) (or n where that sum is zero),
// shapes.h:939-951 + SetShadingGeometry(..., orientationIsAuthoritative = true), interaction.h:194-200
DEV V3 alpha_retrace_origin(V3 p0, V3 p1, V3 p2, float b0, float b1, float b2, bool flip, V3 d, bool smooth = false,
                            V3 n0 = {0, 0, 0}, V3 n1 = {0, 0, 0}, V3 n2 = {0, 0, 0}) {
    constexpr float g7 = gamma_f(7);
    const V3 ph = {(b0 * p0.x + b1 * p1.x) + b2 * p2.x, (b0 * p0.y + b1 * p1.y) + b2 * p2.y,
                   (b0 * p0.z + b1 * p1.z) + b2 * p2.z};
    const V3 pe = {g7 * ((__builtin_fabsf(b0 * p0.x) + __builtin_fabsf(b1 * p1.x)) + __builtin_fabsf(b2 * p2.x)),
                   g7 * ((__builtin_fabsf(b0 * p0.y) + __builtin_fabsf(b1 * p1.y)) + __builtin_fabsf(b2 * p2.y)),
                   g7 * ((__builtin_fabsf(b0 * p0.z) + __builtin_fabsf(b1 * p1.z)) + __builtin_fabsf(b2 * p2.z))};
    // Point3fi(pHit, pError): Interval::FromValueAndError per component (math.h:829-838)
    const V3 lo = {pe.x == 0 ? ph.x : next_down(ph.x + (-pe.x)), pe.y == 0 ? ph.y : next_down(ph.y + (-pe.y)),
                   pe.z == 0 ? ph.z : next_down(ph.z + (-pe.z))};
    const V3 hi = {pe.x == 0 ? ph.x : next_up(ph.x + pe.x), pe.y == 0 ? ph.y : next_up(ph.y + pe.y),
                   pe.z == 0 ? ph.z : next_up(ph.z + pe.z)};
    const V3 c = cross(sub(p0, p2), sub(p1, p2));
    const float len = __builtin_sqrtf(len2(c));  // Normalize: v / Length(v)
    V3 n = {c.x / len, c.y / len, c.z / len};
    if (flip) n = {-n.x, -n.y, -n.z};
    if (smooth) {
        V3 ns = {(b0 * n0.x + b1 * n1.x) + b2 * n2.x, (b0 * n0.y + b1 * n1.y) + b2 * n2.y,
                 (b0 * n0.z + b1 * n1.z) + b2 * n2.z};
        const float l2 = len2(ns);
        if (l2 > 0) {
            const float l = __builtin_sqrtf(l2);
            ns = {ns.x / l, ns.y / l, ns.z / l};
        } else {
            ns = n;
        }
        if (dot_n(n, ns) < 0.f) n = {-n.x, -n.y, -n.z};  // n = FaceForward(n, shading.n)
    }
    return offset_ray_origin(lo, hi, n, d);
}

// ... after an alpha-rejected BILINEAR PATCH hit at (u, v): BilinearPatch::InteractionFromIntersection (shapes.h:
// 1396-1489) for a mesh without (u, v) coordinates gives
//   p = Lerp(u, Lerp(v, p00, p01), Lerp(v, p10, p11)), pError = gamma(6) (|p00| + |p01| + |p10| + |p11|)
//   dpdu = Lerp(v, p10, p11) - Lerp(v, p00, p01), dpdv = Lerp(u, p01, p11) - Lerp(u, p00, p10)
//   n = Normalize(Cross(dpdu, dpdv)), negated under reverseOrientation ^ transformSwapsHandedness
// smooth = the mesh has per-vertex normals: ns = Normalize(bilinear interpolation) where that is not zero, and
// n = FaceForward(n, ns) (SetShadingGeometry, orientationIsAuthoritative = true)
DEV V3 lerp_v3(float t, V3 a, V3 b) {  // (1 - t) * a + t * b: vecmath.h:410-412
    const float s = 1 - t;
    return {s * a.x + t * b.x, s * a.y + t * b.y, s * a.z + t * b.z};
}
// hasUV = the mesh has (u, v) coordinates (uvA = {uv00, uv10}, uvB = {uv01, uv11}): dpdu / dpdv become the
// derivatives with respect to (s, t), shapes.h:1414-1437, before the normal is taken
// `extra` = the slots after the patch's four: {n00} {n10} {n01} {n11} if smooth, then {uv00, uv10} {uv01, uv11} if
// hasUV; they are read where they are needed (the (u, v) pair before the normal, the normals last) so that the
// patch's 20 attribute words are never live together
// uvAt: slot of the (u, v) pair in `extra` (default: right after the normals that are there)
DEV V3 patch_retrace_origin(V3 p00, V3 p10, V3 p01, V3 p11, float u, float v, bool flip, V3 d, bool smooth, bool hasUV,
                            const float4 *extra, int uvAt = -1) {
    const V3 a = lerp_v3(v, p00, p01), b = lerp_v3(v, p10, p11);
    const V3 ph = lerp_v3(u, a, b);
    V3 dpdu = sub(b, a);
    V3 dpdv = sub(lerp_v3(u, p01, p11), lerp_v3(u, p00, p10));
    constexpr float g6 = gamma_f(6);
    const V3 pe = {g6 * (((__builtin_fabsf(p00.x) + __builtin_fabsf(p01.x)) + __builtin_fabsf(p10.x)) + __builtin_fabsf(p11.x)),
                   g6 * (((__builtin_fabsf(p00.y) + __builtin_fabsf(p01.y)) + __builtin_fabsf(p10.y)) + __builtin_fabsf(p11.y)),
                   g6 * (((__builtin_fabsf(p00.z) + __builtin_fabsf(p01.z)) + __builtin_fabsf(p10.z)) + __builtin_fabsf(p11.z))};
    const V3 lo = {pe.x == 0 ? ph.x : next_down(ph.x + (-pe.x)), pe.y == 0 ? ph.y : next_down(ph.y + (-pe.y)),
                   pe.z == 0 ? ph.z : next_down(ph.z + (-pe.z))};
    const V3 hi = {pe.x == 0 ? ph.x : next_up(ph.x + pe.x), pe.y == 0 ? ph.y : next_up(ph.y + pe.y),
                   pe.z == 0 ? ph.z : next_up(ph.z + pe.z)};
    if (hasUV) {
        if (uvAt < 0) uvAt = smooth ? 4 : 0;
        const float4 uvA = extra[uvAt], uvB = extra[uvAt + 1];
        const float sv = 1 - v, su = 1 - u;
        const float s0x = sv * uvA.x + v * uvB.x, s0y = sv * uvA.y + v * uvB.y;  // Lerp(v, uv00, uv01)
        const float s1x = sv * uvA.z + v * uvB.z, s1y = sv * uvA.w + v * uvB.w;  // Lerp(v, uv10, uv11)
        const float dstdu0 = s1x - s0x, dstdu1 = s1y - s0y;
        const float t0x = su * uvB.x + u * uvB.z, t0y = su * uvB.y + u * uvB.w;  // Lerp(u, uv01, uv11)
        const float t1x = su * uvA.x + u * uvA.z, t1y = su * uvA.y + u * uvA.w;  // Lerp(u, uv00, uv10)
        const float dstdv0 = t0x - t1x, dstdv1 = t0y - t1y;
        const float duds = __builtin_fabsf(dstdu0) < 1e-8f ? 0 : 1 / dstdu0;
        const float dvds = __builtin_fabsf(dstdv0) < 1e-8f ? 0 : 1 / dstdv0;
        const float dudt = __builtin_fabsf(dstdu1) < 1e-8f ? 0 : 1 / dstdu1;
        const float dvdt = __builtin_fabsf(dstdv1) < 1e-8f ? 0 : 1 / dstdv1;
        const V3 dpds = {duds * dpdu.x + dvds * dpdv.x, duds * dpdu.y + dvds * dpdv.y, duds * dpdu.z + dvds * dpdv.z};
        V3 dpdt = {dudt * dpdu.x + dvdt * dpdv.x, dudt * dpdu.y + dvdt * dpdv.y, dudt * dpdu.z + dvdt * dpdv.z};
        const V3 c1 = cross(dpds, dpdt);
        if (c1.x != 0 || c1.y != 0 || c1.z != 0) {
            if (dot(cross(dpdu, dpdv), c1) < 0) dpdt = {-dpdt.x, -dpdt.y, -dpdt.z};
            dpdu = dpds;
            dpdv = dpdt;
        }
    }
    const V3 c = cross(dpdu, dpdv);
    const float len = __builtin_sqrtf(len2(c));
    V3 n = {c.x / len, c.y / len, c.z / len};
    if (flip) n = {n.x * -1, n.y * -1, n.z * -1};
    if (smooth) {
        const float4 m0 = extra[0], m1 = extra[1], m2 = extra[2], m3 = extra[3];
        const V3 n00 = {m0.x, m0.y, m0.z}, n10 = {m1.x, m1.y, m1.z}, n01 = {m2.x, m2.y, m2.z}, n11 = {m3.x, m3.y, m3.z};
        const V3 a0 = lerp_v3(v, n00, n01), a1 = lerp_v3(v, n10, n11);
        V3 ns = lerp_v3(u, a0, a1);
        const float l2 = len2(ns);
        if (l2 > 0) {
            const float l = __builtin_sqrtf(l2);
            ns = {ns.x / l, ns.y / l, ns.z / l};
            if (dot_n(n, ns) < 0.f) n = {-n.x, -n.y, -n.z};
        }
    }
    return offset_ray_origin(lo, hi, n, d);
}

}  // namespace nnbvh
