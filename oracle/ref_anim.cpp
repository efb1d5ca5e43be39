// ref_anim.cpp — TEST INFRASTRUCTURE ONLY.
//
// Batch driver around the REFERENCE's own compiled AnimatedTransform (src/pbrt/util/transform.cpp:375-
// 470 constructor with Transform::Decompose, :1062-1081 Interpolate): it contains no restated
// arithmetic.  Built by oracle/Makefile into oracle/_ref/ref_anim with -fno-access-control so that the
// object's private members (T, R, S, actuallyAnimated, hasRotation) can be written out: they are what
// the C ABI's nnbvh_animated_transform carries, and Interpolate(time) is what AnimatedPrimitive::
// Intersect (src/pbrt/cpu/primitive.cpp:140-153) evaluates per ray.
//
// usage: ref_anim <in.bin> <out.bin>
//   in.bin : int32 n, then n records of 35 float32: start m[16] (row-major), end m[16], startTime, endTime, time
//   out.bin: n records of 116 float32: actuallyAnimated, hasRotation, T[0] T[1] (6), R[0] R[1] (8: v.xyz, w),
//            S[0] S[1] (32), start mInv (16), end mInv (16), Interpolate(time).m (16), .mInv (16), pad to 116
#include <pbrt/pbrt.h>
#include <pbrt/util/transform.h>

#include <cstdint>
#include <cstdio>
#include <vector>

using namespace pbrt;

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *fi = std::fopen(argv[1], "rb"), *fo = std::fopen(argv[2], "wb");
    if (!fi || !fo) return 3;
    int32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1) return 4;
    std::vector<float> in((size_t)n * 35);
    if (std::fread(in.data(), 4, in.size(), fi) != in.size()) return 4;
    for (int i = 0; i < n; ++i) {
        const float *r = &in[(size_t)i * 35];
        SquareMatrix<4> ms, me;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                ms[a][b] = r[4 * a + b];
                me[a][b] = r[16 + 4 * a + b];
            }
        Transform ts(ms), te(me);  // Transform(const SquareMatrix<4> &): mInv = Inverse(m)
        AnimatedTransform at(ts, r[32], te, r[33]);
        float out[116] = {0};
        int k = 0;
        out[k++] = at.actuallyAnimated ? 1.f : 0.f;
        out[k++] = at.hasRotation ? 1.f : 0.f;
        for (int j = 0; j < 2; ++j) out[k++] = at.T[j].x, out[k++] = at.T[j].y, out[k++] = at.T[j].z;
        for (int j = 0; j < 2; ++j)
            out[k++] = at.R[j].v.x, out[k++] = at.R[j].v.y, out[k++] = at.R[j].v.z, out[k++] = at.R[j].w;
        for (int j = 0; j < 2; ++j)
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) out[k++] = at.S[j][a][b];
        for (const Transform *t : {&ts, &te})
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) out[k++] = t->GetInverseMatrix()[a][b];
        Transform ti = at.Interpolate(r[34]);
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) out[k++] = ti.GetMatrix()[a][b];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) out[k++] = ti.GetInverseMatrix()[a][b];
        std::fwrite(out, 4, 116, fo);
    }
    std::fclose(fi);
    std::fclose(fo);
    return 0;
}
