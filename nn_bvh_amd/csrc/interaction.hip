// interaction.hip — Triangle:: and BilinearPatch::InteractionFromIntersection on the device
// (/root/reference/src/pbrt/shapes.h:884-1010, 1396-1489, with the SurfaceInteraction constructor
// and SetShadingGeometry they run: interaction.h:32-33, 164-214).
//
// The traversal kernels return what TriangleIntersection carries (primitive, b0 b1 b2, t);
// Triangle::Intersect (shapes.cpp:302-334) then builds the SurfaceInteraction every later stage of
// the reference reads (wavefront/intersect.h:49-156 copies pi, n, dpdu, dpdv, uv, shading.* and
// faceIndex into its work items).  This is that post-pass for a whole batch of hit records: one
// thread per item, ~250 flops, 80-190 B gathered (hit, ray direction + time, 3 vertex indices,
// 3 positions, optional uv / normal / tangent triples) and 192 B written — a streaming, HBM-bound
// pass a few percent of the trace it follows.
//
// Arithmetic is the reference's, operation for operation (FMA exactly where DifferenceOfProducts /
// SumOfProducts / the Normal3 Dot use it, IEEE division and square root, the float vs double
// literal of the two degenerate-uv tests); tests/test_interaction.py checks the kernel bit for
// bit against vectors produced by the compiled reference function.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "interaction.h"
#include "anim_math.h"

namespace nnbvh {

#define IDEV static __device__ __forceinline__

struct F3 {
    float x, y, z;
};
IDEV F3 f3(const float *p) { return {p[0], p[1], p[2]}; }
IDEV F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
IDEV F3 neg(F3 a) { return {-a.x, -a.y, -a.z}; }
IDEV F3 scale(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }  // Tuple3::operator*(U): vecmath.h:350-353
IDEV float len2(F3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }   // vecmath.h:948-950
IDEV float gamma7() { return (7.0f * 0x1p-24f) / (1.0f - 7.0f * 0x1p-24f); }  // float.h:195-197

// dop (DifferenceOfProducts), next_up / next_down: trace_math.h
IDEV float sop(float a, float b, float c, float d) {  // math.h:577-583
    const float cd = c * d;
    const float s = __builtin_fmaf(a, b, cd);
    const float err = __builtin_fmaf(c, d, -cd);
    return s + err;
}
IDEV F3 dop_v(float a, F3 b, float c, F3 d) {  // the same with the vector FMA of vecmath.h:415-417
    return {dop(a, b.x, c, d.x), dop(a, b.y, c, d.y), dop(a, b.z, c, d.z)};
}
IDEV F3 cross(F3 v, F3 w) {  // vecmath.h:932-945, 999-1004
    return {dop(v.y, w.z, v.z, w.y), dop(v.z, w.x, v.x, w.z), dop(v.x, w.y, v.y, w.x)};
}
IDEV float dot_n(F3 n, F3 v) {  // Dot(Normal3, .): vecmath.h:1056-1075
    return __builtin_fmaf(n.x, v.x, sop(n.y, v.y, n.z, v.z));
}
IDEV F3 normalize(F3 v) {  // v / Length(v): vecmath.h:953-961, 362-365
    const float len = __builtin_sqrtf(len2(v));
    return {v.x / len, v.y / len, v.z / len};
}
IDEV void coordinate_system(F3 v1, F3 &v2, F3 &v3) {  // vecmath.h:1007-1013
    const float sign = __builtin_copysignf(1.0f, v1.z);
    const float a = -1 / (sign + v1.z);
    const float b = v1.x * v1.y * a;
    v2 = {1 + sign * (v1.x * v1.x) * a, sign * b, -sign * v1.x};
    v3 = {b, sign + (v1.y * v1.y) * a, -v1.y};
}
IDEV F3 bary(float b0, float b1, float b2, F3 a0, F3 a1, F3 a2) {  // b0 * a0 + b1 * a1 + b2 * a2
    return {(b0 * a0.x + b1 * a1.x) + b2 * a2.x, (b0 * a0.y + b1 * a1.y) + b2 * a2.y,
            (b0 * a0.z + b1 * a1.z) + b2 * a2.z};
}
struct MeshView {
    const float *verts;
    const int32_t *triVerts, *patchVerts;
    const float *normals, *uvs, *tangents;
    const int32_t *faceIndices;
    const uint8_t *triFlags;
    int nTris;
    unsigned defaultFlags;
    const nnbvh_instance *instances;
    int nInstances;
    const float *anim, *animFwd;  // AnimatedPrimitive table + start / end forward rows, or null
};

// Transform::operator()(const SurfaceInteraction &) (util/transform.cpp:229-261) with the instance's
// renderFromPrimitive: what TransformedPrimitive::Intersect does to the hit (cpu/primitive.cpp:122)
IDEV void xf_vec(const float *m, const float *v, float *out) {  // transform.h:322-326
    const float x = v[0], y = v[1], z = v[2];
    for (int i = 0; i < 3; ++i) out[i] = m[4 * i] * x + m[4 * i + 1] * y + m[4 * i + 2] * z;
}
IDEV void xf_normal(const float *mi, const float *n, float *out) {  // transform.h:329-334
    const float x = n[0], y = n[1], z = n[2];
    for (int i = 0; i < 3; ++i) out[i] = mi[i] * x + mi[4 + i] * y + mi[8 + i] * z;
}
IDEV void transform_interaction(const nnbvh_instance &inst, nnbvh_interaction &r) {
    const float *m = inst.render_from_prim, *mi = inst.prim_from_render;
    float x[3], ein[3];
    bool exact = true;
    for (int k = 0; k < 3; ++k) {
        x[k] = (r.pi_lo[k] + r.pi_hi[k]) / 2;
        ein[k] = (r.pi_hi[k] - r.pi_lo[k]) / 2;
        exact = exact && (r.pi_hi[k] - r.pi_lo[k] == 0);
    }
    const float g3 = (3.0f * 0x1p-24f) / (1.0f - 3.0f * 0x1p-24f);
    float lo[3], hi[3];
    for (int i = 0; i < 3; ++i) {
        const float *q = m + 4 * i;
        const float p = (q[0] * x[0] + q[1] * x[1]) + (q[2] * x[2] + q[3]);
        const float a = __builtin_fabsf(q[0] * x[0]) + __builtin_fabsf(q[1] * x[1]) + __builtin_fabsf(q[2] * x[2]) +
                        __builtin_fabsf(q[3]);
        float e;
        if (exact) e = g3 * a;
        else
            e = (g3 + 1) * (__builtin_fabsf(q[0]) * ein[0] + __builtin_fabsf(q[1]) * ein[1] +
                            __builtin_fabsf(q[2]) * ein[2]) + g3 * a;
        if (e == 0) {
            lo[i] = hi[i] = p;
        } else {
            lo[i] = next_down(p + (-e));
            hi[i] = next_up(p + e);
        }
    }
    for (int k = 0; k < 3; ++k) r.pi_lo[k] = lo[k], r.pi_hi[k] = hi[k];
    float t[3];
    xf_normal(mi, r.n, t);
    F3 n = normalize(F3{t[0], t[1], t[2]});
    xf_vec(m, r.wo, t);
    const F3 wo = normalize(F3{t[0], t[1], t[2]});
    r.wo[0] = wo.x, r.wo[1] = wo.y, r.wo[2] = wo.z;
    xf_vec(m, r.dpdu, t);
    r.dpdu[0] = t[0], r.dpdu[1] = t[1], r.dpdu[2] = t[2];
    xf_vec(m, r.dpdv, t);
    r.dpdv[0] = t[0], r.dpdv[1] = t[1], r.dpdv[2] = t[2];
    xf_normal(mi, r.dndu, t);
    r.dndu[0] = t[0], r.dndu[1] = t[1], r.dndu[2] = t[2];
    xf_normal(mi, r.dndv, t);
    r.dndv[0] = t[0], r.dndv[1] = t[1], r.dndv[2] = t[2];
    xf_normal(mi, r.ns, t);
    F3 ns = normalize(F3{t[0], t[1], t[2]});
    xf_vec(m, r.dpdus, t);
    r.dpdus[0] = t[0], r.dpdus[1] = t[1], r.dpdus[2] = t[2];
    xf_vec(m, r.dpdvs, t);
    r.dpdvs[0] = t[0], r.dpdvs[1] = t[1], r.dpdvs[2] = t[2];
    xf_normal(mi, r.dndus, t);
    r.dndus[0] = t[0], r.dndus[1] = t[1], r.dndus[2] = t[2];
    xf_normal(mi, r.dndvs, t);
    r.dndvs[0] = t[0], r.dndvs[1] = t[1], r.dndvs[2] = t[2];
    if (dot_n(ns, n) < 0.f) ns = neg(ns);  // shading.n = FaceForward(shading.n, n) (:257)
    r.n[0] = n.x, r.n[1] = n.y, r.n[2] = n.z;
    r.ns[0] = ns.x, r.ns[1] = ns.y, r.ns[2] = ns.z;
}

IDEV F3 lerp3(float t, F3 a, F3 b) {  // (1 - t) * a + t * b: vecmath.h:410-412
    const float s = 1 - t;
    return {s * a.x + t * b.x, s * a.y + t * b.y, s * a.z + t * b.z};
}
IDEV F3 scale_add2(F3 a, float sa, F3 b, float sb) {  // a * sa + b * sb
    return {sa * a.x + sb * b.x, sa * a.y + sb * b.y, sa * a.z + sb * b.z};
}
IDEV float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vecmath.h:964-967
IDEV float comp(F3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
IDEV void put(float *dst, F3 v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

// BilinearPatch::InteractionFromIntersection (shapes.h:1396-1489), with RotateFromTo
// (util/transform.h:249-270) for the shading frame
IDEV void patch_interaction(const MeshView &m, int prim, float u, float v, F3 wo, float time, nnbvh_interaction &r) {
    const int v0 = m.patchVerts[4 * (long)prim], v1 = m.patchVerts[4 * (long)prim + 1],
              v2 = m.patchVerts[4 * (long)prim + 2], v3 = m.patchVerts[4 * (long)prim + 3];
    const unsigned flags = m.triFlags ? m.triFlags[prim] : m.defaultFlags;
    const F3 p00 = f3(m.verts + 3 * (long)v0), p10 = f3(m.verts + 3 * (long)v1), p01 = f3(m.verts + 3 * (long)v2),
             p11 = f3(m.verts + 3 * (long)v3);
    const F3 a = lerp3(v, p00, p01), b = lerp3(v, p10, p11);
    const F3 p = lerp3(u, a, b);
    F3 dpdu = b - a;
    F3 dpdv = lerp3(u, p01, p11) - lerp3(u, p00, p10);
    float st0 = u, st1 = v;
    float duds = 1, dudt = 0, dvds = 0, dvdt = 1;
    if ((flags & NNBVH_TRI_HAS_UV) && m.uvs) {
        const float *q00 = m.uvs + 2 * (long)v0, *q10 = m.uvs + 2 * (long)v1, *q01 = m.uvs + 2 * (long)v2,
                    *q11 = m.uvs + 2 * (long)v3;
        const float sv = 1 - v, su = 1 - u;
        const float s0x = sv * q00[0] + v * q01[0], s0y = sv * q00[1] + v * q01[1];
        const float s1x = sv * q10[0] + v * q11[0], s1y = sv * q10[1] + v * q11[1];
        st0 = su * s0x + u * s1x;
        st1 = su * s0y + u * s1y;
        const float dstdu0 = s1x - s0x, dstdu1 = s1y - s0y;
        const float t0x = su * q01[0] + u * q11[0], t0y = su * q01[1] + u * q11[1];
        const float t1x = su * q00[0] + u * q10[0], t1y = su * q00[1] + u * q10[1];
        const float dstdv0 = t0x - t1x, dstdv1 = t0y - t1y;
        duds = __builtin_fabsf(dstdu0) < 1e-8f ? 0 : 1 / dstdu0;
        dvds = __builtin_fabsf(dstdv0) < 1e-8f ? 0 : 1 / dstdv0;
        dudt = __builtin_fabsf(dstdu1) < 1e-8f ? 0 : 1 / dstdu1;
        dvdt = __builtin_fabsf(dstdv1) < 1e-8f ? 0 : 1 / dstdv1;
        const F3 dpds = scale_add2(dpdu, duds, dpdv, dvds);
        F3 dpdt = scale_add2(dpdu, dudt, dpdv, dvdt);
        const F3 c1 = cross(dpds, dpdt);
        if (c1.x != 0 || c1.y != 0 || c1.z != 0) {
            if (dot3(cross(dpdu, dpdv), c1) < 0) dpdt = neg(dpdt);
            dpdu = dpds;
            dpdv = dpdt;
        }
    }
    // fundamental forms (:1441-1456); d2Pduu = d2Pdvv = 0
    const F3 d2uv = {(p00.x - p01.x) + (p11.x - p10.x), (p00.y - p01.y) + (p11.y - p10.y),
                     (p00.z - p01.z) + (p11.z - p10.z)};
    const F3 zero = {0, 0, 0};
    const float E = dot3(dpdu, dpdu), F = dot3(dpdu, dpdv), G = dot3(dpdv, dpdv);
    const F3 nn = normalize(cross(dpdu, dpdv));
    const float e = dot3(nn, zero), f = dot3(nn, d2uv), g = dot3(nn, zero);
    const float EGF2 = dop(E, G, F, F);
    const float invEGF2 = (EGF2 == 0) ? 0.0f : 1 / EGF2;
    F3 dndu = scale_add2(dpdu, (f * F - e * G) * invEGF2, dpdv, (e * F - f * E) * invEGF2);
    F3 dndv = scale_add2(dpdu, (g * F - f * G) * invEGF2, dpdv, (f * F - g * E) * invEGF2);
    const F3 dnds = scale_add2(dndu, duds, dndv, dvds), dndt = scale_add2(dndu, dudt, dndv, dvdt);
    dndu = dnds;
    dndv = dndt;
    const float g6 = (6.0f * 0x1p-24f) / (1.0f - 6.0f * 0x1p-24f);
    const float pe[3] = {
        g6 * (((__builtin_fabsf(p00.x) + __builtin_fabsf(p01.x)) + __builtin_fabsf(p10.x)) + __builtin_fabsf(p11.x)),
        g6 * (((__builtin_fabsf(p00.y) + __builtin_fabsf(p01.y)) + __builtin_fabsf(p10.y)) + __builtin_fabsf(p11.y)),
        g6 * (((__builtin_fabsf(p00.z) + __builtin_fabsf(p01.z)) + __builtin_fabsf(p10.z)) + __builtin_fabsf(p11.z))};
    // SurfaceInteraction(pi, st, wo, dpdu, dpdv, dndu, dndv, time, flipNormal): interaction.h:164-183
    F3 nrm = normalize(cross(dpdu, dpdv));
    if (flags & NNBVH_TRI_FLIP_NORMAL) nrm = {nrm.x * -1, nrm.y * -1, nrm.z * -1};
    F3 ns = nrm, sdpdu = dpdu, sdpdv = dpdv, sdndu = dndu, sdndv = dndv;
    if ((flags & NNBVH_TRI_HAS_N) && m.normals) {
        const F3 n00 = f3(m.normals + 3 * (long)v0), n10 = f3(m.normals + 3 * (long)v1),
                 n01 = f3(m.normals + 3 * (long)v2), n11 = f3(m.normals + 3 * (long)v3);
        const F3 a0 = lerp3(v, n00, n01), a1 = lerp3(v, n10, n11);
        const F3 nsv = lerp3(u, a0, a1);
        if (len2(nsv) > 0) {
            const F3 nsn = normalize(nsv);
            const F3 du = a1 - a0;
            const F3 dv = lerp3(u, n01, n11) - lerp3(u, n00, n10);
            const F3 ds = scale_add2(du, duds, dv, dvds), dt = scale_add2(du, dudt, dv, dvdt);
            const F3 from = normalize(nrm);
            F3 refl = {0, 0, 0};
            if (__builtin_fabsf(from.x) < 0.72f && __builtin_fabsf(nsn.x) < 0.72f) refl.x = 1;
            else if (__builtin_fabsf(from.y) < 0.72f && __builtin_fabsf(nsn.y) < 0.72f) refl.y = 1;
            else refl.z = 1;
            const F3 uu = refl - from, vv = refl - nsn;
            const float duu = dot3(uu, uu), dvv = dot3(vv, vv), duv = dot3(uu, vv);
            float rd[3], re[3];
            for (int i = 0; i < 3; ++i) {
                float row[3];
                for (int j = 0; j < 3; ++j)
                    row[j] = ((i == j) ? 1 : 0) - 2 / duu * comp(uu, i) * comp(uu, j) -
                             2 / dvv * comp(vv, i) * comp(vv, j) + 4 * duv / (duu * dvv) * comp(vv, i) * comp(uu, j);
                rd[i] = row[0] * dpdu.x + row[1] * dpdu.y + row[2] * dpdu.z;
                re[i] = row[0] * dpdv.x + row[1] * dpdv.y + row[2] * dpdv.z;
            }
            // SetShadingGeometry(ns, r(dpdu), r(dpdv), dndu, dndv, true)
            ns = nsn;
            if (dot_n(nrm, ns) < 0.f) nrm = neg(nrm);
            sdpdu = {rd[0], rd[1], rd[2]};
            sdpdv = {re[0], re[1], re[2]};
            sdndu = ds;
            sdndv = dt;
            while (len2(sdpdu) > 1e16f || len2(sdpdv) > 1e16f) {
                sdpdu = {sdpdu.x / 1e8f, sdpdu.y / 1e8f, sdpdu.z / 1e8f};
                sdpdv = {sdpdv.x / 1e8f, sdpdv.y / 1e8f, sdpdv.z / 1e8f};
            }
        }
    }
    const float ph[3] = {p.x, p.y, p.z};
    for (int k = 0; k < 3; ++k) {
        if (pe[k] == 0) {
            r.pi_lo[k] = r.pi_hi[k] = ph[k];
        } else {
            r.pi_lo[k] = next_down(ph[k] + (-pe[k]));
            r.pi_hi[k] = next_up(ph[k] + pe[k]);
        }
    }
    r.uv[0] = st0, r.uv[1] = st1;
    put(r.wo, normalize(wo));
    r.time = time;
    put(r.n, nrm);
    r.face_index = m.faceIndices ? m.faceIndices[prim] : 0;
    put(r.dpdu, dpdu);
    put(r.dpdv, dpdv);
    put(r.ns, ns);
    put(r.dpdus, sdpdu);
    put(r.dpdvs, sdpdv);
    put(r.dndus, sdndu);
    put(r.dndvs, sdndv);
    put(r.dndu, dndu);
    put(r.dndv, dndv);
}

// FULL = false: the mesh has neither bilinear patches nor an instance table — the patch interaction, the
// instance / AnimatedPrimitive transforms and their registers are compiled out (crown: 1.00 -> see DESIGN.md §5.6)
#ifndef NNBVH_INTR_LEAN_WAVES
#define NNBVH_INTR_LEAN_WAVES 4
#endif
template <bool FULL>
__global__ __launch_bounds__(256, FULL ? 1 : NNBVH_INTR_LEAN_WAVES) void k_triangle_interactions(
    MeshView m, const float4 *__restrict__ rays, nnbvh_ray_soa soa, const float4 *__restrict__ hits, int n,
    const int32_t *nDev, nnbvh_interaction *__restrict__ out) {
    if (nDev) {
        const int nd = *nDev;
        n = nd < 0 ? 0 : (nd < n ? nd : n);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 h0 = hits[2 * (long)i], h1 = hits[2 * (long)i + 1];
        const int prim = __float_as_int(h0.x);
        nnbvh_interaction r;
        __builtin_memset(&r, 0, sizeof r);
        r.prim = prim;
        r.status = NNBVH_INTERACTION_MISS;
        if (prim >= 0) {
            r.status = NNBVH_INTERACTION_HOST;
            const int inst = __float_as_int(h1.w);  // 0 top level, k + 1 inside instance k
            if ((inst == 0 || (inst > 0 && inst <= m.nInstances)) && prim < m.nTris) {
                if (m.triVerts[3 * (long)prim] >= 0) r.status = NNBVH_INTERACTION_TRIANGLE;
                else if (FULL && m.patchVerts && m.patchVerts[4 * (long)prim] >= 0) r.status = NNBVH_INTERACTION_PATCH;
            }
        }
        if (r.status != NNBVH_INTERACTION_TRIANGLE && r.status != NNBVH_INTERACTION_PATCH) {
            // only prim / status are meaningful: store the record's last 16 B {prim, status, 0, 0}
            reinterpret_cast<float4 *>(out + i)[11] =
                make_float4(__int_as_float(prim), __int_as_float(r.status), 0.0f, 0.0f);
            continue;
        }
        F3 wo;
        float time;
        if (rays) {
            const float4 r1 = rays[2 * (long)i + 1];
            wo = {-r1.x, -r1.y, -r1.z};  // Triangle::Intersect passes -ray.d (shapes.cpp:331)
            time = r1.w;
        } else {
            wo = {-soa.dx[i], -soa.dy[i], -soa.dz[i]};
            time = soa.time ? soa.time[i] : 0.0f;
        }
        const int instIdx = FULL ? __float_as_int(h1.w) - 1 : -1;
        nnbvh_instance xf;  // the instance's transform as this ray sees it
        if (FULL && instIdx >= 0) {
            xf = m.instances[instIdx];
            if (m.anim && m.anim[(long)kAnimStride * instIdx + 74] != 0.0f) {
                // AnimatedPrimitive::Intersect (cpu/primitive.cpp:143-153): renderFromPrimitive.Interpolate(r.time),
                // both for the ray into the instance's space and for the interaction back out of it
                float4 r0, r1, r2, f0, f1, f2;
                anim_rows<true>(m.anim + (long)kAnimStride * instIdx, m.animFwd + 24l * instIdx, time, r0, r1, r2, f0, f1, f2);
                const float fr[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
                const float ir[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                for (int k = 0; k < 12; ++k) xf.render_from_prim[k] = fr[k], xf.prim_from_render[k] = ir[k];
            }
            // TransformedPrimitive::Intersect (cpu/primitive.cpp:112-125): the shape sees the ray in the
            // instance's space, ray.d = renderFromPrimitive.ApplyInverse(r.d) (util/transform.h:401-405)
            const float *mi = xf.prim_from_render;
            const F3 d = neg(wo);
            const F3 di = {mi[0] * d.x + mi[1] * d.y + mi[2] * d.z, mi[4] * d.x + mi[5] * d.y + mi[6] * d.z,
                           mi[8] * d.x + mi[9] * d.y + mi[10] * d.z};
            wo = neg(di);
        }
        if (FULL && r.status == NNBVH_INTERACTION_PATCH) {
            patch_interaction(m, prim, h0.z, h0.w, wo, time, r);
            if (instIdx >= 0) transform_interaction(xf, r);
            out[i] = r;
            continue;
        }
        const float b0 = h0.z, b1 = h0.w, b2 = h1.x;
        const int v0 = m.triVerts[3 * (long)prim], v1 = m.triVerts[3 * (long)prim + 1],
                  v2 = m.triVerts[3 * (long)prim + 2];
        const unsigned flags = m.triFlags ? m.triFlags[prim] : m.defaultFlags;
        const F3 p0 = f3(m.verts + 3 * (long)v0), p1 = f3(m.verts + 3 * (long)v1), p2 = f3(m.verts + 3 * (long)v2);
        float uv[6] = {0, 0, 1, 0, 1, 1};  // shapes.h:898
        if ((flags & NNBVH_TRI_HAS_UV) && m.uvs) {
            uv[0] = m.uvs[2 * (long)v0], uv[1] = m.uvs[2 * (long)v0 + 1];
            uv[2] = m.uvs[2 * (long)v1], uv[3] = m.uvs[2 * (long)v1 + 1];
            uv[4] = m.uvs[2 * (long)v2], uv[5] = m.uvs[2 * (long)v2 + 1];
        }
        const float duv02x = uv[0] - uv[4], duv02y = uv[1] - uv[5];
        const float duv12x = uv[2] - uv[4], duv12y = uv[3] - uv[5];
        const F3 dp02 = p0 - p2, dp12 = p1 - p2;
        const float determinant = dop(duv02x, duv12y, duv02y, duv12x);
        F3 dpdu = {0, 0, 0}, dpdv = {0, 0, 0};
        const bool degenerateUV = __builtin_fabsf(determinant) < 1e-9f;
        if (!degenerateUV) {
            const float invdet = 1 / determinant;
            dpdu = scale(invdet, dop_v(duv12y, dp02, duv02y, dp12));
            dpdv = scale(invdet, dop_v(duv02x, dp12, duv12x, dp02));
        }
        if (degenerateUV || len2(cross(dpdu, dpdv)) == 0) {
            const F3 e20 = p2 - p0, e10 = p1 - p0;
            F3 ng = cross(e20, e10);
            if (len2(ng) == 0) {  // shapes.h:916-919: the cross product again, in double
                const double vx = e20.x, vy = e20.y, vz = e20.z, wx = e10.x, wy = e10.y, wz = e10.z;
                auto dopd = [](double a, double b, double c, double d) {
                    const double cd = c * d;
                    const double diff = __builtin_fma(a, b, -cd);
                    const double err = __builtin_fma(-c, d, cd);
                    return diff + err;
                };
                ng = {(float)dopd(vy, wz, vz, wy), (float)dopd(vz, wx, vx, wz), (float)dopd(vx, wy, vy, wx)};
                // the reference CHECK-aborts on a zero normal here; IntersectTriangle never reports
                // such a triangle as hit, so the record is simply marked for the host
                if (len2(ng) == 0) {
                    reinterpret_cast<float4 *>(out + i)[11] =
                        make_float4(__int_as_float(prim), __int_as_float(NNBVH_INTERACTION_HOST), 0.0f, 0.0f);
                    continue;
                }
            }
            coordinate_system(normalize(ng), dpdu, dpdv);
        }
        const F3 pHit = bary(b0, b1, b2, p0, p1, p2);
        const float uvHitU = (b0 * uv[0] + b1 * uv[2]) + b2 * uv[4];
        const float uvHitV = (b0 * uv[1] + b1 * uv[3]) + b2 * uv[5];
        const float g7 = gamma7();
        const float pex = g7 * ((__builtin_fabsf(b0 * p0.x) + __builtin_fabsf(b1 * p1.x)) + __builtin_fabsf(b2 * p2.x));
        const float pey = g7 * ((__builtin_fabsf(b0 * p0.y) + __builtin_fabsf(b1 * p1.y)) + __builtin_fabsf(b2 * p2.y));
        const float pez = g7 * ((__builtin_fabsf(b0 * p0.z) + __builtin_fabsf(b1 * p1.z)) + __builtin_fabsf(b2 * p2.z));
        // isect.n = isect.shading.n = Normalize(Cross(dp02, dp12)), flipped by orientation (:933-936)
        F3 nrm = normalize(cross(dp02, dp12));
        if (flags & NNBVH_TRI_FLIP_NORMAL) nrm = neg(nrm);
        F3 ns = nrm, sdpdu = dpdu, sdpdv = dpdv, dndu = {0, 0, 0}, dndv = {0, 0, 0};
        const bool hasN = (flags & NNBVH_TRI_HAS_N) && m.normals, hasS = (flags & NNBVH_TRI_HAS_S) && m.tangents;
        if (hasN || hasS) {
            F3 n0 = {0, 0, 0}, n1 = n0, n2 = n0;
            F3 nsv = nrm;
            if (hasN) {
                n0 = f3(m.normals + 3 * (long)v0), n1 = f3(m.normals + 3 * (long)v1), n2 = f3(m.normals + 3 * (long)v2);
                const F3 t = bary(b0, b1, b2, n0, n1, n2);
                if (len2(t) > 0) nsv = normalize(t);
            }
            F3 ss = dpdu;
            if (hasS) {
                const F3 t = bary(b0, b1, b2, f3(m.tangents + 3 * (long)v0), f3(m.tangents + 3 * (long)v1),
                                  f3(m.tangents + 3 * (long)v2));
                if (len2(t) != 0) ss = t;
            }
            F3 ts = cross(nsv, ss);
            if (len2(ts) > 0) ss = cross(ts, nsv);
            else coordinate_system(nsv, ss, ts);
            if (hasN) {
                const F3 dn1 = n0 - n2, dn2 = n1 - n2;
                const float det2 = dop(duv02x, duv12y, duv02y, duv12x);
                if ((double)__builtin_fabsf(det2) < 1e-9) {  // :963 compares against a double literal
                    const F3 dn = cross(n2 - n0, n1 - n0);
                    if (len2(dn) != 0) coordinate_system(dn, dndu, dndv);
                } else {
                    const float invDet = 1 / det2;
                    dndu = scale(invDet, dop_v(duv12y, dn1, duv02y, dn2));
                    dndv = scale(invDet, dop_v(duv02x, dn2, duv12x, dn1));
                }
            }
            // SetShadingGeometry(ns, ss, ts, dndu, dndv, true): interaction.h:194-214
            ns = nsv;
            if (dot_n(nrm, ns) < 0.f) nrm = neg(nrm);
            sdpdu = ss;
            sdpdv = ts;
            while (len2(sdpdu) > 1e16f || len2(sdpdv) > 1e16f) {
                sdpdu = {sdpdu.x / 1e8f, sdpdu.y / 1e8f, sdpdu.z / 1e8f};
                sdpdv = {sdpdv.x / 1e8f, sdpdv.y / 1e8f, sdpdv.z / 1e8f};
            }
        }
        const float ph[3] = {pHit.x, pHit.y, pHit.z}, pe[3] = {pex, pey, pez};
        for (int k = 0; k < 3; ++k) {  // Point3fi(pHit, pError): vecmath.h:751-754
            if (pe[k] == 0) {
                r.pi_lo[k] = r.pi_hi[k] = ph[k];
            } else {
                r.pi_lo[k] = next_down(ph[k] + (-pe[k]));
                r.pi_hi[k] = next_up(ph[k] + pe[k]);
            }
        }
        const F3 won = normalize(wo);  // Interaction(): wo(Normalize(wo)), interaction.h:32-33
        r.uv[0] = uvHitU, r.uv[1] = uvHitV;
        r.wo[0] = won.x, r.wo[1] = won.y, r.wo[2] = won.z;
        r.time = time;
        r.n[0] = nrm.x, r.n[1] = nrm.y, r.n[2] = nrm.z;
        r.face_index = m.faceIndices ? m.faceIndices[prim] : 0;
        r.dpdu[0] = dpdu.x, r.dpdu[1] = dpdu.y, r.dpdu[2] = dpdu.z;
        r.dpdv[0] = dpdv.x, r.dpdv[1] = dpdv.y, r.dpdv[2] = dpdv.z;
        r.ns[0] = ns.x, r.ns[1] = ns.y, r.ns[2] = ns.z;
        r.dpdus[0] = sdpdu.x, r.dpdus[1] = sdpdu.y, r.dpdus[2] = sdpdu.z;
        r.dpdvs[0] = sdpdv.x, r.dpdvs[1] = sdpdv.y, r.dpdvs[2] = sdpdv.z;
        r.dndus[0] = dndu.x, r.dndus[1] = dndu.y, r.dndus[2] = dndu.z;
        r.dndvs[0] = dndv.x, r.dndvs[1] = dndv.y, r.dndvs[2] = dndv.z;
        if (instIdx >= 0) transform_interaction(xf, r);
#ifdef NNBVH_INTR_PROBE_NOWRITE  // tools only: what the 192-B record stores cost (the record is folded into 16 B)
        {
            const float *f = reinterpret_cast<const float *>(&r);
            float4 acc = {0, 0, 0, 0};
            for (int k = 0; k < 48; k += 4) acc = {acc.x + f[k], acc.y + f[k + 1], acc.z + f[k + 2], acc.w + f[k + 3]};
            reinterpret_cast<float4 *>(out + i)[11] = acc;
        }
#else
        out[i] = r;
#endif
    }
}

hipError_t launch_triangle_interactions(const ShadingMeshDevice &m, const void *rays, const nnbvh_ray_soa *soa,
                                        const void *hits, int n, const int32_t *nDev, void *out, int maxBlocks,
                                        hipStream_t stream) {
    MeshView v{m.verts, m.triVerts, m.patchVerts, m.normals, m.uvs, m.tangents, m.faceIndices, m.triFlags, m.nTris, m.defaultFlags,
               m.instances, m.nInstances, m.anim, m.animFwd};
    nnbvh_ray_soa s;
    __builtin_memset(&s, 0, sizeof s);
    if (soa) s = *soa;
    int blocks = (n + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks < maxBlocks ? blocks : maxBlocks);
    if (m.patchVerts || m.instances)
        hipLaunchKernelGGL(k_triangle_interactions<true>, dim3(blocks), dim3(256), 0, stream, v, (const float4 *)rays, s,
                           (const float4 *)hits, n, nDev, (nnbvh_interaction *)out);
    else
        hipLaunchKernelGGL(k_triangle_interactions<false>, dim3(blocks), dim3(256), 0, stream, v, (const float4 *)rays, s,
                           (const float4 *)hits, n, nDev, (nnbvh_interaction *)out);
    return hipGetLastError();
}

}  // namespace nnbvh
