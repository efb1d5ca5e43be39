// nnbvh_aggregate.hpp — header-only C++17 adapter over the C ABI (include/nnbvh.h) with the
// shape of pbrt's aggregate interfaces, so a pbrt build can use the HIP path unchanged:
//
//   per-ray     Primitive::{Bounds, Intersect, IntersectP}
//               (/root/reference/src/pbrt/cpu/primitive.h:33-45; BVHAggregate's versions at
//                cpu/aggregates.h:28-70, cpu/aggregates.cpp:524-624)
//   batched     WavefrontAggregate::{Bounds, IntersectClosest, IntersectShadow}
//               (/root/reference/src/pbrt/wavefront/integrator.h:32-54; CPU implementation
//                wavefront/aggregate.cpp:34-68)
//
// The class owns only the opaque scene handle.  Error behaviour mirrors the reference: pbrt
// aborts through CHECK / LOG_FATAL (util/check.h:36-57, cpu/aggregates.cpp:145); here every
// failure goes through nnbvh::HipBVHAggregate::fatal, which prints the C ABI's message and
// aborts unless the embedder installs its own handler (e.g. one that calls pbrt's ErrorExit).
//
// A per-ray Intersect() is a batch of one: correct, and three orders of magnitude slower
// than the batched calls (a kernel launch per ray).  It exists so that the CPU integrators
// can call the aggregate "unchanged" for validation; production callers batch.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <optional>
#include <string>
#include <vector>

#include "nnbvh.h"

namespace nnbvh {

struct Point3f {
    float x, y, z;
};
using Vector3f = Point3f;
struct Bounds3f {
    Point3f pMin, pMax;
};
// the fields of pbrt's Ray the path reads (ray.h:18-39)
struct Ray {
    Point3f o;
    Vector3f d;
    float time = 0;
};
// what TriangleIntersection / BilinearIntersection + the primitive index carry
// (shapes.h:820-824, 1271-1275); a pbrt embedder turns this into a ShapeIntersection with
// Triangle::InteractionFromIntersection (shapes.h:884-1010) on the host.
struct HitRecord {
    int prim;
    float tHit;
    float b0, b1, b2;  // patch: b0 = u, b1 = v
};

class HipBVHAggregate {
  public:
    using FatalHandler = void (*)(const char *);
    static FatalHandler &fatal_handler() {
        static FatalHandler h = [](const char *msg) {
            std::fprintf(stderr, "nnbvh: fatal: %s\n", msg);
            std::abort();
        };
        return h;
    }
    static void fatal(const std::string &what) {
        fatal_handler()((what + ": " + nnbvh_last_error()).c_str());
    }

    // == BVHAggregate::Create / ctor (aggregates.cpp:725-744, 140-190): builds on the host.
    // splitMethod: "sah" (default) | "middle" | "equal"; maxPrimsInNode default 4.
    HipBVHAggregate(const std::vector<nnbvh_prim> &prims, const std::vector<float> &verts,
                    int maxPrimsInNode = 4, const std::string &splitMethod = "sah", int device = 0) {
        int method = splitMethod == "sah"      ? NNBVH_SPLIT_SAH
                     : splitMethod == "middle" ? NNBVH_SPLIT_MIDDLE
                     : splitMethod == "equal"  ? NNBVH_SPLIT_EQUAL_COUNTS
                     : splitMethod == "hlbvh"  ? NNBVH_SPLIT_HLBVH
                                               : -1;
        nnbvh_build *b = nnbvh_build_create(prims.data(), (int)prims.size(), verts.data(),
                                            (int)(verts.size() / 3), maxPrimsInNode, method);
        if (!b) {
            fatal("HipBVHAggregate: build");
            return;
        }
        int nNodes = 0, nPrims = 0;
        const nnbvh_linear_node *nodes = nnbvh_build_nodes(b, &nNodes);
        const nnbvh_prim *ordered = nnbvh_build_ordered_prims(b, &nPrims);
        scene_ = nnbvh_scene_create(nodes, nNodes, ordered, nPrims, verts.data(),
                                    (int)(verts.size() / 3), device);
        nnbvh_build_destroy(b);
        if (!scene_) fatal("HipBVHAggregate: scene_create");
    }

    // from a tree pbrt itself built: BVHAggregate::nodes + the leaf-ordered primitives
    HipBVHAggregate(const nnbvh_linear_node *nodes, int nNodes, const nnbvh_prim *orderedPrims,
                    int nPrims, const float *verts, int nVerts, int device = 0) {
        scene_ = nnbvh_scene_create(nodes, nNodes, orderedPrims, nPrims, verts, nVerts, device);
        if (!scene_) fatal("HipBVHAggregate: scene_create");
    }

    HipBVHAggregate(const HipBVHAggregate &) = delete;
    HipBVHAggregate &operator=(const HipBVHAggregate &) = delete;
    ~HipBVHAggregate() { nnbvh_scene_destroy(scene_); }

    // ---- Primitive interface -----------------------------------------------------------
    Bounds3f Bounds() const {
        float b[6];
        if (nnbvh_scene_bounds(scene_, b) != NNBVH_OK) fatal("Bounds");
        return {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
    }

    std::optional<HitRecord> Intersect(const Ray &ray,
                                       float tMax = std::numeric_limits<float>::infinity()) const {
        nnbvh_ray r = wire(ray, tMax);
        nnbvh_hit h;
        if (nnbvh_intersect_closest(scene_, &r, 1, &h) != NNBVH_OK) fatal("Intersect");
        if (h.prim < 0) return {};
        return HitRecord{h.prim, h.t, h.b0, h.b1, h.b2};
    }

    bool IntersectP(const Ray &ray, float tMax = std::numeric_limits<float>::infinity()) const {
        nnbvh_ray r = wire(ray, tMax);
        uint8_t occ = 0;
        if (nnbvh_intersect_any(scene_, &r, 1, &occ, nullptr, nullptr) != NNBVH_OK)
            fatal("IntersectP");
        return occ != 0;
    }

    // ---- WavefrontAggregate-shaped batches (host buffers; synchronous) -------------------
    void IntersectClosest(const nnbvh_ray *rays, int64_t n, nnbvh_hit *hits) const {
        if (nnbvh_intersect_closest(scene_, rays, n, hits) != NNBVH_OK) fatal("IntersectClosest");
    }
    void IntersectShadow(const nnbvh_ray *rays, int64_t n, uint8_t *occluded,
                         int32_t *nodesVisited = nullptr, int32_t *primTests = nullptr) const {
        if (nnbvh_intersect_any(scene_, rays, n, occluded, nodesVisited, primTests) != NNBVH_OK)
            fatal("IntersectShadow");
    }

    // ---- device-resident batches, stream-ordered (hipStream_t as void*) ------------------
    void IntersectClosestDevice(const void *dRays, int64_t n, void *dHits, void *stream) const {
        if (nnbvh_intersect_closest_device(scene_, dRays, n, dHits, stream) != NNBVH_OK)
            fatal("IntersectClosestDevice");
    }
    void IntersectShadowDevice(const void *dRays, int64_t n, void *dOccluded, void *stream,
                               void *dNodesVisited = nullptr, void *dPrimTests = nullptr) const {
        if (nnbvh_intersect_any_device(scene_, dRays, n, dOccluded, dNodesVisited, dPrimTests,
                                       stream) != NNBVH_OK)
            fatal("IntersectShadowDevice");
    }

    // ---- whole wavefront stages on the device: the SOA queue in, the reference's destination
    //      queues (as index queues) and pixel radiance out; see include/nnbvh.h --------------
    // == WavefrontAggregate::IntersectClosest (wavefront/integrator.h:37-43)
    void IntersectClosestQueues(int maxRays, const nnbvh_ray_soa &rayQueue, const int32_t *dSize,
                                const uint8_t *dPrimClass, int64_t nPrimClass, void *dHits,
                                const nnbvh_closest_queues &out, void *stream) const {
        if (nnbvh_wavefront_intersect_closest(scene_, maxRays, &rayQueue, dSize, dPrimClass,
                                              nPrimClass, dHits, &out, stream) != NNBVH_OK)
            fatal("IntersectClosestQueues");
    }
    // == WavefrontAggregate::IntersectShadow (wavefront/integrator.h:45-46)
    void IntersectShadowQueue(int maxRays, const nnbvh_ray_soa &shadowQueue, const int32_t *dSize,
                              const float *dLd, const float *dRu, const float *dRl,
                              const int32_t *dPixelIndex, float *dL, int64_t nPixels,
                              void *stream, uint8_t *dOccluded = nullptr) const {
        if (nnbvh_wavefront_intersect_shadow(scene_, maxRays, &shadowQueue, dSize, dLd, dRu, dRl,
                                             dPixelIndex, dL, nPixels, dOccluded, stream) != NNBVH_OK)
            fatal("IntersectShadowQueue");
    }

    nnbvh_scene *handle() const { return scene_; }

  private:
    static nnbvh_ray wire(const Ray &ray, float tMax) {
        return nnbvh_ray{{ray.o.x, ray.o.y, ray.o.z}, tMax, {ray.d.x, ray.d.y, ray.d.z}, ray.time};
    }
    nnbvh_scene *scene_ = nullptr;
};

// Triangle:: / BilinearPatch::InteractionFromIntersection (shapes.h:884-1010, 1396-1489) for batches of hit records: the mesh
// side of a pbrt scene (TriangleMesh arrays flattened over all meshes) + the post-pass.
class HipShadingMesh {
  public:
    HipShadingMesh(const float *verts, int nVerts, const int32_t *triVertices, int nPrims,
                   const float *normals = nullptr, const float *uvs = nullptr,
                   const float *tangents = nullptr, const int32_t *faceIndices = nullptr,
                   const uint8_t *triFlags = nullptr, int device = 0,
                   const int32_t *patchVertices = nullptr)
        : mesh_(nnbvh_shading_mesh_create(verts, nVerts, triVertices, patchVertices, nPrims, normals,
                                          uvs, tangents, faceIndices, triFlags, device)) {
        if (!mesh_) HipBVHAggregate::fatal("HipShadingMesh");
    }
    ~HipShadingMesh() { nnbvh_shading_mesh_destroy(mesh_); }
    HipShadingMesh(const HipShadingMesh &) = delete;
    HipShadingMesh &operator=(const HipShadingMesh &) = delete;

    void Interactions(const nnbvh_ray *rays, const nnbvh_hit *hits, int32_t n, nnbvh_interaction *out) const {
        if (nnbvh_triangle_interactions(mesh_, rays, hits, n, out) != NNBVH_OK)
            HipBVHAggregate::fatal("Interactions");
    }
    // rays either as records (dRays) or as the wavefront SOA queue (raySoa); stream-ordered
    void InteractionsDevice(const void *dRays, const nnbvh_ray_soa *raySoa, const void *dHits,
                            int32_t maxItems, const int32_t *dSize, void *dOut, void *stream) const {
        if (nnbvh_triangle_interactions_device(mesh_, dRays, raySoa, dHits, maxItems, dSize, dOut,
                                               stream) != NNBVH_OK)
            HipBVHAggregate::fatal("InteractionsDevice");
    }

  private:
    nnbvh_shading_mesh *mesh_ = nullptr;
};

}  // namespace nnbvh
