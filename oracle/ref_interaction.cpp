// ref_interaction.cpp — TEST INFRASTRUCTURE ONLY.
//
// Batch driver around the REFERENCE's own Triangle::InteractionFromIntersection
// (/root/reference/src/pbrt/shapes.h:884-1010, inline) on a one-triangle TriangleMesh built by
// the reference's own constructor (src/pbrt/util/mesh.cpp:23-77, identity transform).  No restated
// arithmetic.  Built by oracle/Makefile into oracle/_ref/ref_interaction (git-ignored); used to
// validate oracle/nnbvh_oracle.c's orc_triangle_interaction and to generate
// tests/golden/tri_interaction.npz (tests/golden/make_interaction_golden.py).
//
// With mode "blp": BilinearPatch::InteractionFromIntersection (shapes.h:1396-1489) on a one-patch
// BilinearPatchMesh built by the reference's constructor (util/mesh.cpp:183-232).
//   in.bin : int32 n, then n records of 40 float32: p00 p10 p01 p11 [0:12], hit (u,v) [12:14],
//            wo [14:17], time [17], flags [18] (1 = mesh has uv, 2 = has n, 8 = reverseOrientation),
//            uv00 uv10 uv01 uv11 [19:27], n00 n10 n01 n11 [27:39], pad
//   out.bin: n records of 50 float32: the 38 below, pi low[3] / high[3], geometric dndu[3] dndv[3]
//
// With mode "xf": Transform::operator()(const SurfaceInteraction &) (util/transform.cpp:229-261), what
// TransformedPrimitive::Intersect applies to a hit inside an instance (cpu/primitive.cpp:112-125).
//   in.bin : int32 n, then n records of 72 float32: m[16] mInv[16] (row-major), pi low[3] high[3],
//            n, wo, dpdu, dpdv, dndu, dndv, shading n, dpdu, dpdv, dndu, dndv (3 each), pad
//   out.bin: n records of 40 float32: pi low[3] high[3] and the same eleven vectors, pad
//
// usage: ref_interaction [tri|blp|xf] <in.bin> <out.bin>      (mode defaults to tri)
//   in.bin : int32 n, then n records of 36 float32:
//            p0[3] p1[3] p2[3]  b0 b1 b2  wo[3]  time  flags  uv0[2] uv1[2] uv2[2]  n0[3] n1[3] n2[3]  s0? -> see below
//            flags bit 0: mesh has uv, bit 1: mesh has n, bit 2: mesh has s, bit 3: reverseOrientation
//            (36 floats = 9 + 3 + 3 + 1 + 1 + 6 + 9 + 4 pad; tangents s0..s2 follow as 9 more = 45)
//   out.bin: n records of 38 float32: p[3] pError[3] uv[2] wo[3] n[3] dpdu[3] dpdv[3]
//            ns[3] dpdus[3] dpdvs[3] dndus[3] dndvs[3] time faceIndex(as float)
#include <pbrt/pbrt.h>
#include <pbrt/interaction.h>
#include <pbrt/options.h>
#include <pbrt/shapes.h>
#include <pbrt/util/buffercache.h>
#include <pbrt/util/mesh.h>
#include <pbrt/util/transform.h>
#include <pbrt/util/vecmath.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

using namespace pbrt;

static constexpr int kIn = 45, kOut = 38;

static void put_common(const SurfaceInteraction &si, float *out) {
    Point3f ph = si.p();
    Vector3f pe = si.pi.Error();
    float v[38] = {ph.x, ph.y, ph.z, pe.x, pe.y, pe.z, si.uv[0], si.uv[1], si.wo.x, si.wo.y, si.wo.z,
                   si.n.x, si.n.y, si.n.z, si.dpdu.x, si.dpdu.y, si.dpdu.z, si.dpdv.x, si.dpdv.y, si.dpdv.z,
                   si.shading.n.x, si.shading.n.y, si.shading.n.z,
                   si.shading.dpdu.x, si.shading.dpdu.y, si.shading.dpdu.z,
                   si.shading.dpdv.x, si.shading.dpdv.y, si.shading.dpdv.z,
                   si.shading.dndu.x, si.shading.dndu.y, si.shading.dndu.z,
                   si.shading.dndv.x, si.shading.dndv.y, si.shading.dndv.z,
                   si.time, (float)si.faceIndex, 0.f};
    for (int k = 0; k < 38; ++k) out[k] = v[k];
    out[38] = si.pi.x.LowerBound(), out[39] = si.pi.y.LowerBound(), out[40] = si.pi.z.LowerBound();
    out[41] = si.pi.x.UpperBound(), out[42] = si.pi.y.UpperBound(), out[43] = si.pi.z.UpperBound();
}

static int run_patches(FILE *fi, FILE *fo) {
    constexpr int kInB = 40, kOutB = 50;
    int32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1) return 4;
    std::vector<float> in((size_t)n * kInB);
    if (std::fread(in.data(), 4, in.size(), fi) != in.size()) return 4;
    Allocator alloc;
    for (int i = 0; i < n; ++i) {
        const float *r = &in[(size_t)i * kInB];
        const int flags = (int)r[18];
        std::vector<Point3f> p = {Point3f(r[0], r[1], r[2]), Point3f(r[3], r[4], r[5]), Point3f(r[6], r[7], r[8]),
                                  Point3f(r[9], r[10], r[11])};
        std::vector<Point2f> uv;
        std::vector<Normal3f> N;
        if (flags & 1)
            uv = {Point2f(r[19], r[20]), Point2f(r[21], r[22]), Point2f(r[23], r[24]), Point2f(r[25], r[26])};
        if (flags & 2)
            N = {Normal3f(r[27], r[28], r[29]), Normal3f(r[30], r[31], r[32]), Normal3f(r[33], r[34], r[35]),
                 Normal3f(r[36], r[37], r[38])};
        BilinearPatchMesh mesh(Transform(), (flags & 8) != 0, {0, 1, 2, 3}, p, N, uv, {7 + i}, nullptr, alloc);
        SurfaceInteraction si = BilinearPatch::InteractionFromIntersection(&mesh, 0, Point2f(r[12], r[13]), r[17],
                                                                           Vector3f(r[14], r[15], r[16]));
        float out[kOutB];
        put_common(si, out);
        out[44] = si.dndu.x, out[45] = si.dndu.y, out[46] = si.dndu.z;
        out[47] = si.dndv.x, out[48] = si.dndv.y, out[49] = si.dndv.z;
        std::fwrite(out, 4, kOutB, fo);
    }
    return 0;
}

static int run_transform(FILE *fi, FILE *fo) {
    constexpr int kInX = 72, kOutX = 40;
    int32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1) return 4;
    std::vector<float> in((size_t)n * kInX);
    if (std::fread(in.data(), 4, in.size(), fi) != in.size()) return 4;
    for (int i = 0; i < n; ++i) {
        const float *r = &in[(size_t)i * kInX];
        SquareMatrix<4> m, mInv;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                m[a][b] = r[4 * a + b];
                mInv[a][b] = r[16 + 4 * a + b];
            }
        Transform xf(m, mInv);
        SurfaceInteraction si;
        si.pi = Point3fi(Interval(r[32], r[35]), Interval(r[33], r[36]), Interval(r[34], r[37]));
        auto V = [&](int at) { return Vector3f(r[at], r[at + 1], r[at + 2]); };
        auto N = [&](int at) { return Normal3f(r[at], r[at + 1], r[at + 2]); };
        si.n = N(38), si.wo = V(41), si.dpdu = V(44), si.dpdv = V(47), si.dndu = N(50), si.dndv = N(53);
        si.shading.n = N(56), si.shading.dpdu = V(59), si.shading.dpdv = V(62);
        si.shading.dndu = N(65), si.shading.dndv = N(68);
        SurfaceInteraction t = xf(si);
        float out[kOutX] = {t.pi.x.LowerBound(), t.pi.y.LowerBound(), t.pi.z.LowerBound(),
                            t.pi.x.UpperBound(), t.pi.y.UpperBound(), t.pi.z.UpperBound(),
                            t.n.x, t.n.y, t.n.z, t.wo.x, t.wo.y, t.wo.z, t.dpdu.x, t.dpdu.y, t.dpdu.z,
                            t.dpdv.x, t.dpdv.y, t.dpdv.z, t.dndu.x, t.dndu.y, t.dndu.z, t.dndv.x, t.dndv.y, t.dndv.z,
                            t.shading.n.x, t.shading.n.y, t.shading.n.z,
                            t.shading.dpdu.x, t.shading.dpdu.y, t.shading.dpdu.z,
                            t.shading.dpdv.x, t.shading.dpdv.y, t.shading.dpdv.z,
                            t.shading.dndu.x, t.shading.dndu.y, t.shading.dndu.z,
                            t.shading.dndv.x, t.shading.dndv.y, t.shading.dndv.z, 0.f};
        std::fwrite(out, 4, kOutX, fo);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3 && argc != 4) return 2;
    const bool patches = argc == 4 && std::string(argv[1]) == "blp";
    if (argc == 4 && std::string(argv[1]) == "xf") {
        FILE *fi = std::fopen(argv[2], "rb"), *fo = std::fopen(argv[3], "wb");
        if (!fi || !fo) return 3;
        int rc = run_transform(fi, fo);
        std::fclose(fo);
        return rc;
    }
    FILE *fi = std::fopen(argv[argc - 2], "rb");
    FILE *fo = std::fopen(argv[argc - 1], "wb");
    if (!fi || !fo) return 3;
    Options = new PBRTOptions;
    InitBufferCaches();
    if (patches) {
        int rc = run_patches(fi, fo);
        std::fclose(fo);
        return rc;
    }
    int32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1) return 4;
    std::vector<float> in((size_t)n * kIn);
    if (std::fread(in.data(), 4, in.size(), fi) != in.size()) return 4;
    Allocator alloc;
    for (int i = 0; i < n; ++i) {
        const float *r = &in[(size_t)i * kIn];
        const int flags = (int)r[19];
        std::vector<Point3f> p = {Point3f(r[0], r[1], r[2]), Point3f(r[3], r[4], r[5]), Point3f(r[6], r[7], r[8])};
        std::vector<Point2f> uv;
        std::vector<Normal3f> N;
        std::vector<Vector3f> S;
        if (flags & 1) uv = {Point2f(r[20], r[21]), Point2f(r[22], r[23]), Point2f(r[24], r[25])};
        if (flags & 2) N = {Normal3f(r[26], r[27], r[28]), Normal3f(r[29], r[30], r[31]), Normal3f(r[32], r[33], r[34])};
        if (flags & 4) S = {Vector3f(r[36], r[37], r[38]), Vector3f(r[39], r[40], r[41]), Vector3f(r[42], r[43], r[44])};
        TriangleMesh mesh(Transform(), (flags & 8) != 0, {0, 1, 2}, p, S, N, uv, {7 + i}, alloc);
        TriangleIntersection ti{r[9], r[10], r[11], 1.f};
        SurfaceInteraction si = Triangle::InteractionFromIntersection(&mesh, 0, ti, r[18], Vector3f(r[12], r[13], r[14]));
        Point3f ph = si.p();
        Vector3f pe = si.pi.Error();
        float out[kOut] = {ph.x, ph.y, ph.z, pe.x, pe.y, pe.z, si.uv[0], si.uv[1], si.wo.x, si.wo.y, si.wo.z,
                           si.n.x, si.n.y, si.n.z, si.dpdu.x, si.dpdu.y, si.dpdu.z, si.dpdv.x, si.dpdv.y, si.dpdv.z,
                           si.shading.n.x, si.shading.n.y, si.shading.n.z,
                           si.shading.dpdu.x, si.shading.dpdu.y, si.shading.dpdu.z,
                           si.shading.dpdv.x, si.shading.dpdv.y, si.shading.dpdv.z,
                           si.shading.dndu.x, si.shading.dndu.y, si.shading.dndu.z,
                           si.shading.dndv.x, si.shading.dndv.y, si.shading.dndv.z,
                           si.time, (float)si.faceIndex, 0.f};
        // the exact interval as well: low/high of pi replace p/pError when asked for by tests
        std::fwrite(out, 4, kOut, fo);
        float iv[6] = {si.pi.x.LowerBound(), si.pi.y.LowerBound(), si.pi.z.LowerBound(),
                       si.pi.x.UpperBound(), si.pi.y.UpperBound(), si.pi.z.UpperBound()};
        std::fwrite(iv, 4, 6, fo);
    }
    std::fclose(fo);
    return 0;
}
