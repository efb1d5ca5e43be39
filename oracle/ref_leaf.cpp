// ref_leaf.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin batch driver around the REFERENCE's own compiled leaf functions.  It contains
// no restated arithmetic: it includes the reference headers where they lie under
// /root/reference/src and calls
//   pbrt::IntersectTriangle          (src/pbrt/shapes.cpp:172-273; object compiled from that file)
//   pbrt::IntersectBilinearPatch     (src/pbrt/shapes.h:1279-1347, inline)
//   pbrt::Bounds3f::IntersectP       (src/pbrt/util/vecmath.h:1573-1608, inline; 7-argument
//                                     traversal overload, with invDir/dirIsNeg prepared as
//                                     src/pbrt/cpu/aggregates.cpp:534-535 does)
// Built by oracle/Makefile into oracle/_ref/ref_leaf (git-ignored).  Used to validate
// oracle/nnbvh_oracle.c and to generate tests/golden/leaf_*.bin
// (tests/golden/make_leaf_golden.py).  Never shipped, never on the product path.
//
//   pbrt::Transform::ApplyInverse(Ray, Float *tMax)   (src/pbrt/util/transform.h:416-429, inline:
//                                     the interval-arithmetic ray transform TransformedPrimitive
//                                     applies, src/pbrt/cpu/primitive.cpp:112-131)
//   pbrt::Transform::operator()(const Bounds3f &)     (src/pbrt/util/transform.cpp:134-139: the
//                                     bounds TransformedPrimitive::Bounds() reports, cpu/primitive.h:91)
//   pbrt::EncodeMorton3 + Bounds3f::Offset            (src/pbrt/util/math.h:99-119, util/vecmath.h:1322-1331:
//                                     the Morton code buildHLBVH gives a primitive, cpu/aggregates.cpp:398-408)
//   pbrt::Bounds3f::IntersectP(o, d, tMax, &t0, &t1)  (src/pbrt/util/vecmath.h:1547-1571, inline: the
//                                     interval KdTreeAggregate::Intersect starts from, cpu/aggregates.cpp:975)
//   pbrt::Hash / HashFloat                           (src/pbrt/util/hash.h:19-128, inline: the stochastic alpha
//                                     test of GeometricPrimitive::Intersect, cpu/primitive.cpp:62, and the
//                                     seed of IntersectOneRandom, wavefront/aggregate.cpp:94)
//   pbrt::OffsetRayOrigin / SpawnRayTo               (src/pbrt/ray.h:75-101, inline)
//   pbrt::WeightedReservoirSampler + RNG             (src/pbrt/util/sampling.h:524-596, util/rng.h, inline)
// usage: ref_leaf <tri|blp|slab|xfray|xfbounds|morton|slab2|hash|offset|wrs> <in.bin> <out.bin>
//   in.bin : int32 n, then n records of float32
//              tri : o[3] d[3] tmax p0[3] p1[3] p2[3]              (16 floats)
//              blp : o[3] d[3] tmax p00[3] p10[3] p01[3] p11[3]    (19 floats)
//              slab: o[3] d[3] tmax pmin[3] pmax[3]                (13 floats)
//              xfray: o[3] d[3] tmax m[16] mInv[16] (row-major)     (39 floats)
//              xfbounds: m[16] (row-major) pmin[3] pmax[3]          (22 floats)
//              morton: centroid-bounds pmin[3] pmax[3], centroid[3]  (9 floats)
//              slab2: as slab                                        (13 floats)
//              hash: o[3] d[3]                                       (6 floats)
//              offset: pi_lo[3] pi_hi[3] n[3] w[3]                   (12 floats; w = direction and target point)
//              wrs: p0[3] p1[3] nAdds                                (7 floats)
//   out.bin: n records: tri  -> int32 hit, float b0 b1 b2 t
//                       blp  -> int32 hit, float u v t
//                       slab -> int32 hit
//                       xfray-> int32 1, float o'[3] d'[3] tmax'
//                       xfbounds-> int32 1, float pmin'[3] pmax'[3]
//                       morton-> int32 code
//                       slab2-> int32 hit, float t0 t1
//                       hash -> int32 low 32 bits of Hash(o, d), float HashFloat(o, d), high 32 bits (bit pattern)
//                       offset-> int32 1, float OffsetRayOrigin(pi, n, w)[3], SpawnRayTo(pi, n, 0, w) o[3] d[3]
//                       wrs  -> int32 selected index (-1 = no sample), float SampleProbability, WeightSum
#include <pbrt/pbrt.h>
#include <pbrt/ray.h>
#include <pbrt/shapes.h>
#include <pbrt/util/hash.h>
#include <pbrt/util/rng.h>
#include <pbrt/util/sampling.h>
#include <pbrt/util/transform.h>
#include <pbrt/util/vecmath.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace pbrt;

static Point3f P(const float *p) { return Point3f(p[0], p[1], p[2]); }

int main(int argc, char **argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: ref_leaf <tri|blp|slab> in.bin out.bin\n");
        return 2;
    }
    int mode = !std::strcmp(argv[1], "tri") ? 0 : !std::strcmp(argv[1], "blp") ? 1
               : !std::strcmp(argv[1], "slab") ? 2 : !std::strcmp(argv[1], "xfray") ? 3
               : !std::strcmp(argv[1], "xfbounds") ? 4 : !std::strcmp(argv[1], "morton") ? 5
               : !std::strcmp(argv[1], "slab2") ? 6 : !std::strcmp(argv[1], "hash") ? 7
               : !std::strcmp(argv[1], "offset") ? 8 : !std::strcmp(argv[1], "wrs") ? 9 : -1;
    if (mode < 0) return 2;
    const int stride[10] = {16, 19, 13, 39, 22, 9, 13, 6, 12, 7};
    FILE *fi = std::fopen(argv[2], "rb");
    FILE *fo = std::fopen(argv[3], "wb");
    if (!fi || !fo) return 3;
    int32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1) return 4;
    std::vector<float> in((size_t)n * stride[mode]);
    if (std::fread(in.data(), 4, in.size(), fi) != in.size()) return 4;
    for (int i = 0; i < n; ++i) {
        const float *r = &in[(size_t)i * stride[mode]];
        Ray ray(P(r), Vector3f(r[3], r[4], r[5]));
        float tMax = r[6];
        int32_t hit = 0;
        float out[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int nout = 0;
        if (mode == 0) {
            auto ti = IntersectTriangle(ray, tMax, P(r + 7), P(r + 10), P(r + 13));
            nout = 4;
            if (ti) {
                hit = 1;
                out[0] = ti->b0, out[1] = ti->b1, out[2] = ti->b2, out[3] = ti->t;
            }
        } else if (mode == 1) {
            auto bi = IntersectBilinearPatch(ray, tMax, P(r + 7), P(r + 10), P(r + 13), P(r + 16));
            nout = 3;
            if (bi) {
                hit = 1;
                out[0] = bi->uv[0], out[1] = bi->uv[1], out[2] = bi->t;
            }
        } else if (mode == 3) {
            SquareMatrix<4> m, mInv;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    m[a][b] = r[7 + 4 * a + b];
                    mInv[a][b] = r[23 + 4 * a + b];
                }
            Transform xf(m, mInv);
            Float t = tMax;
            Ray tr = xf.ApplyInverse(ray, &t);
            hit = 1;
            nout = 7;
            out[0] = tr.o.x, out[1] = tr.o.y, out[2] = tr.o.z;
            out[3] = tr.d.x, out[4] = tr.d.y, out[5] = tr.d.z;
            out[6] = t;
        } else if (mode == 6) {
            Bounds3f b;
            b.pMin = P(r + 7);
            b.pMax = P(r + 10);
            Float t0 = 0, t1 = 0;
            hit = b.IntersectP(ray.o, ray.d, tMax, &t0, &t1) ? 1 : 0;
            nout = 2;
            if (hit) out[0] = t0, out[1] = t1;
        } else if (mode == 7) {
            const uint64_t h = Hash(ray.o, ray.d);
            hit = (int32_t)(uint32_t)h;
            nout = 2;
            out[0] = HashFloat(ray.o, ray.d);
            const uint32_t hi = (uint32_t)(h >> 32);
            std::memcpy(&out[1], &hi, 4);
        } else if (mode == 8) {
            Point3fi pi(Interval(r[0], r[3]), Interval(r[1], r[4]), Interval(r[2], r[5]));
            Normal3f n(r[6], r[7], r[8]);
            Vector3f w(r[9], r[10], r[11]);
            Point3f po = OffsetRayOrigin(pi, n, w);
            Ray sr = SpawnRayTo(pi, n, 0.f, Point3f(r[9], r[10], r[11]));
            hit = 1;
            nout = 9;
            out[0] = po.x, out[1] = po.y, out[2] = po.z;
            out[3] = sr.o.x, out[4] = sr.o.y, out[5] = sr.o.z;
            out[6] = sr.d.x, out[7] = sr.d.y, out[8] = sr.d.z;
        } else if (mode == 9) {
            // the sampler of CPUAggregate::IntersectOneRandom (wavefront/aggregate.cpp:94-107): seeded by
            // Hash(p0, p1), every candidate added with weight 1
            WeightedReservoirSampler<int> wrs(Hash(P(r), P(r + 3)));
            const int nAdds = (int)r[6];
            for (int k = 0; k < nAdds; ++k) wrs.Add(k, 1.f);
            hit = wrs.HasSample() ? wrs.GetSample() : -1;
            nout = 2;
            out[0] = wrs.HasSample() ? wrs.SampleProbability() : 0.f;
            out[1] = wrs.WeightSum();
        } else if (mode == 5) {
            // exactly the statements of buildHLBVH (cpu/aggregates.cpp:398-408)
            Bounds3f bounds(P(r), P(r + 3));
            constexpr int mortonBits = 10;
            constexpr int mortonScale = 1 << mortonBits;
            Vector3f centroidOffset = bounds.Offset(P(r + 6));
            Vector3f offset = centroidOffset * mortonScale;
            hit = (int32_t)EncodeMorton3(offset.x, offset.y, offset.z);
        } else if (mode == 4) {
            SquareMatrix<4> m;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) m[a][b] = r[4 * a + b];
            Transform xf(m);
            Bounds3f b;
            b.pMin = P(r + 16);
            b.pMax = P(r + 19);
            Bounds3f bt = xf(b);
            hit = 1;
            nout = 6;
            out[0] = bt.pMin.x, out[1] = bt.pMin.y, out[2] = bt.pMin.z;
            out[3] = bt.pMax.x, out[4] = bt.pMax.y, out[5] = bt.pMax.z;
        } else {
            Bounds3f b;
            b.pMin = P(r + 7);
            b.pMax = P(r + 10);
            Vector3f invDir(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
            int dirIsNeg[3] = {int(invDir.x < 0), int(invDir.y < 0), int(invDir.z < 0)};
            hit = b.IntersectP(ray.o, ray.d, tMax, invDir, dirIsNeg) ? 1 : 0;
        }
        std::fwrite(&hit, 4, 1, fo);
        if (nout) std::fwrite(out, 4, nout, fo);
    }
    std::fclose(fi);
    std::fclose(fo);
    return 0;
}
