// nnbvh_internal.h — shared between the host-side pieces of libnnbvh_hip.so.
#pragma once
#include <cstdint>
#include <string>

#include "../../include/nnbvh.h"

namespace nnbvh {

void set_error(const std::string &msg);

// ---- device data layout -----------------------------------
// One 64-byte record per INTERIOR node, holding both children's boxes, so that one
// coalesced 64-B fetch decides both child visits (the reference does two dependent 32-B
// LinearBVHNode fetches for the same decisions).  Records are numbered in the order
// the interior nodes appear in the reference's DFS array.
//   q0 = { c0.pmin.x, c0.pmin.y, c0.pmin.z, c0.pmax.x }
//   q1 = { c0.pmax.y, c0.pmax.z, c1.pmin.x, c1.pmin.y }
//   q2 = { c1.pmin.z, c1.pmax.x, c1.pmax.y, c1.pmax.z }
//   q3 = { ref0, ref1, axis, 0 }   (int32 bit patterns)
// c0 = the node at index+1 (first child), c1 = the node at secondChildOffset.
// ref >= 0: interior record number; ref < 0: leaf, ~ref = first 16-B slot of its
// primitives in the prim stream.
struct WideNode {
    float q[12];
    int32_t ref0, ref1;
    int32_t axis;
    int32_t pad;
};
static_assert(sizeof(WideNode) == 64, "WideNode must be 64 bytes");

// Prim stream: 16-B slots, primitives in the reference's leaf order.
//   triangle (3 slots): {p0.xyz, id} {p1.xyz, flags} {p2.xyz, 0}
//   patch    (4 slots): {p00.xyz, id} {p10.xyz, flags} {p01.xyz, 0} {p11.xyz, 0}
// flags bit0 = last primitive of its leaf, bit1 = bilinear patch, bit2 = degenerate triangle
// (LengthSquared(Cross(p2 - p0, p1 - p0)) == 0, shapes.cpp:176-177, evaluated at bake time).
constexpr uint32_t kPrimLast = 1u;
constexpr uint32_t kPrimPatch = 2u;
constexpr uint32_t kPrimDegenerate = 4u;
// instance (TransformedPrimitive) record, 6 slots:
//   {childRoot.pmin.xyz, instance index} {childRoot.pmax.xyz, flags}
//   {mInv row 0} {mInv row 1} {mInv row 2}  (renderFromPrimitive inverse, 3x4)
//   {child root ref, 0, 0, 0}
constexpr uint32_t kPrimInstance = 8u;
// host-only primitive (3 slots, no geometry): {0,0,0,id} {0,0,0,flags} {0,0,0,0}
constexpr uint32_t kPrimHost = 16u;

// alpha-tested triangle (GeometricPrimitive with a constant alpha, cpu/primitive.cpp:57-70): slot 2's
// fourth word holds alpha; kPrimFlipN = mesh->reverseOrientation ^ mesh->transformSwapsHandedness
constexpr uint32_t kPrimAlpha = 32u;
constexpr uint32_t kPrimFlipN = 64u;
// instance record of an AnimatedPrimitive: the inverse matrix is interpolated per ray from the
// animation table (anim_math.h) instead of read from slots 2..4
constexpr uint32_t kPrimAnimated = 128u;
// alpha-tested bilinear patch (kPrimPatch | kPrimAlpha): slot 2's fourth word holds alpha like a triangle's; with
// kPrimSmooth four more slots {n00,0} {n10,0} {n01,0} {n11,0} follow the patch's four; with kPrimUV two more,
// {uv00, uv10} {uv01, uv11}, after those
// alpha-tested triangle of a mesh WITH per-vertex shading normals: three more slots {n0,0} {n1,0} {n2,0} follow
// (the re-trace after a rejected hit offsets along FaceForward(n, ns), shapes.h:939-951)
constexpr uint32_t kPrimSmooth = 256u;
constexpr uint32_t kPrimUV = 512u;  // alpha-tested patch of a mesh with (u, v) coordinates
constexpr int kAnimStride = 76;  // floats per entry of the animation table (layout: anim_math.h)
inline bool is_triangle_kind(int kind) {
    return kind == NNBVH_PRIM_TRIANGLE || kind == NNBVH_PRIM_ALPHA_TRIANGLE || kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED ||
           kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH || kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED;
}
inline bool is_smooth_alpha_kind(int kind) {
    return kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH || kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED;
}
// alpha-tested bilinear patches: kind = 8 + flipped + 2 * smooth + 4 * uv
constexpr bool is_alpha_patch_kind(int kind) { return kind >= NNBVH_PRIM_ALPHA_PATCH && kind <= NNBVH_PRIM_ALPHA_PATCH_UV_SMOOTH_FLIPPED; }
constexpr bool is_smooth_alpha_patch_kind(int kind) { return is_alpha_patch_kind(kind) && ((kind - NNBVH_PRIM_ALPHA_PATCH) & 2); }
constexpr bool is_flipped_alpha_patch_kind(int kind) { return is_alpha_patch_kind(kind) && ((kind - NNBVH_PRIM_ALPHA_PATCH) & 1); }
constexpr bool is_uv_alpha_patch_kind(int kind) { return is_alpha_patch_kind(kind) && ((kind - NNBVH_PRIM_ALPHA_PATCH) & 4); }
constexpr int alpha_patch_slots(int kind) { return 4 + (is_smooth_alpha_patch_kind(kind) ? 4 : 0) + (is_uv_alpha_patch_kind(kind) ? 2 : 0); }
inline bool is_flat_alpha_kind(int kind) {
    return kind == NNBVH_PRIM_ALPHA_TRIANGLE || kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED;
}

constexpr int kMaxStack = 64;  // the reference's nodesToVisit[64], aggregates.cpp:538

}  // namespace nnbvh
