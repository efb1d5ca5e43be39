// nnbvh_aggregate.hpp — header-only C++17 adapter over the C ABI (include/nnbvh.h) with the
// shape of pbrt's aggregate interfaces, so a pbrt build can use the HIP path unchanged:
//
//   per-ray     Primitive::{Bounds, Intersect, IntersectP}
//               (/root/reference/src/pbrt/cpu/primitive.h:33-45; BVHAggregate's versions at
//                cpu/aggregates.h:28-70, cpu/aggregates.cpp:524-624)
//   batched     WavefrontAggregate::{Bounds, IntersectClosest, IntersectShadow, IntersectShadowTr,
//               IntersectOneRandom}
//               (/root/reference/src/pbrt/wavefront/integrator.h:32-54; CPU implementation
//                wavefront/aggregate.cpp:34-116; the last two in their media-free form)
//   kd-tree     KdTreeAggregate (cpu/aggregates.h:75-105) as HipKdTreeAggregate
//   film        RGBFilm's accumulators (film.h:232-316) as HipFilm
//
// The class owns only the opaque scene handle.  Error behaviour mirrors the reference: pbrt
// aborts through CHECK / LOG_FATAL (util/check.h:36-57, cpu/aggregates.cpp:145); here every
// failure goes through nnbvh::HipBVHAggregate::fatal, which prints the C ABI's message and
// aborts unless the embedder installs its own handler (e.g. one that calls pbrt's ErrorExit).
//
// A per-ray Intersect() is a batch of one: correct, and three orders of magnitude slower
// than the batched calls (a kernel launch per ray).  It exists so that the CPU integrators
// can call the aggregate "unchanged" for validation; production callers batch.
// CoalescingAggregate<> is the middle way for those integrators: the rays of the many threads of
// pbrt's ParallelFor2D (util/parallel.cpp:301-330) that are inside Intersect() at the same moment are
// traced as ONE batch.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
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
    int instance = 0;  // 0 = top level, k + 1 = inside instance k (nnbvh_hit.instance)
};
// A ray whose result is VOID: a primitive only the host can intersect (NNBVH_PRIM_HOST: quadric, curve,
// textured alpha ...) lay on its way, so neither "hit", "miss" nor "occluded" is known and the caller must
// re-trace it on the CPU (nnbvh_hit.instance == -1 / occluded == 2).  The single-ray adapters report it
// through the optional `needsHost` out-parameter; without one such a ray is fatal, never a silent answer.

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
    // primBounds: 6 floats (min, max) per primitive, read for NNBVH_PRIM_HOST / NNBVH_PRIM_INSTANCE entries
    // (what Primitive::Bounds() returns for them); may be null when the list holds neither
    // normals: the meshes' per-vertex normals (3 floats per vertex), read for the smooth alpha-tested kinds;
    // primAlpha: one constant alpha per entry of `prims`, read for the alpha-tested bilinear patches; uvs: per-vertex
    // (u, v), read for the alpha-tested patches of meshes with uv
    HipBVHAggregate(const std::vector<nnbvh_prim> &prims, const std::vector<float> &verts,
                    int maxPrimsInNode = 4, const std::string &splitMethod = "sah", int device = 0,
                    const std::vector<float> *primBounds = nullptr, const std::vector<float> *normals = nullptr,
                    const std::vector<float> *primAlpha = nullptr, const std::vector<float> *uvs = nullptr) {
        int method = splitMethod == "sah"      ? NNBVH_SPLIT_SAH
                     : splitMethod == "middle" ? NNBVH_SPLIT_MIDDLE
                     : splitMethod == "equal"  ? NNBVH_SPLIT_EQUAL_COUNTS
                     : splitMethod == "hlbvh"  ? NNBVH_SPLIT_HLBVH
                                               : -1;
        // the builder reorders the primitives; with a per-primitive array to carry along, build with the
        // position as the id and put the caller's ids back afterwards
        std::vector<nnbvh_prim> tagged;
        if (primAlpha) {
            tagged = prims;
            for (size_t i = 0; i < tagged.size(); ++i) tagged[i].id = (int32_t)i;
        }
        const nnbvh_prim *in = primAlpha ? tagged.data() : prims.data();
        nnbvh_build *b = primBounds
                             ? nnbvh_build_create_with_bounds(in, (int)prims.size(), verts.data(),
                                                              (int)(verts.size() / 3), primBounds->data(),
                                                              maxPrimsInNode, method)
                             : nnbvh_build_create(in, (int)prims.size(), verts.data(),
                                                  (int)(verts.size() / 3), maxPrimsInNode, method);
        if (!b) {
            fatal("HipBVHAggregate: build");
            return;
        }
        int nNodes = 0, nPrims = 0;
        const nnbvh_linear_node *nodes = nnbvh_build_nodes(b, &nNodes);
        const nnbvh_prim *ordered = nnbvh_build_ordered_prims(b, &nPrims);
        std::vector<nnbvh_prim> restored;
        std::vector<float> alphaOrdered;
        if (primAlpha) {
            restored.assign(ordered, ordered + nPrims);
            alphaOrdered.resize((size_t)nPrims);
            for (int i = 0; i < nPrims; ++i) {
                const size_t from = (size_t)restored[(size_t)i].id;
                alphaOrdered[(size_t)i] = (*primAlpha)[from];
                restored[(size_t)i].id = prims[from].id;
            }
            ordered = restored.data();
        }
        scene_ = (normals || primAlpha || uvs)
                     ? nnbvh_scene_create_with_attributes(nodes, nNodes, ordered, nPrims, verts.data(),
                                                          normals ? normals->data() : nullptr,
                                                          uvs ? uvs->data() : nullptr,
                                                          primAlpha ? alphaOrdered.data() : nullptr,
                                                          (int)(verts.size() / 3), device)
                     : nnbvh_scene_create(nodes, nNodes, ordered, nPrims, verts.data(), (int)(verts.size() / 3), device);
        nnbvh_build_destroy(b);
        if (!scene_) fatal("HipBVHAggregate: scene_create");
    }

    // from a tree pbrt itself built: BVHAggregate::nodes + the leaf-ordered primitives
    // (normals as above; primAlpha indexed like orderedPrims)
    HipBVHAggregate(const nnbvh_linear_node *nodes, int nNodes, const nnbvh_prim *orderedPrims,
                    int nPrims, const float *verts, int nVerts, int device = 0, const float *normals = nullptr,
                    const float *primAlpha = nullptr, const float *uvs = nullptr) {
        scene_ = (normals || primAlpha || uvs)
                     ? nnbvh_scene_create_with_attributes(nodes, nNodes, orderedPrims, nPrims, verts, normals, uvs,
                                                          primAlpha, nVerts, device)
                     : nnbvh_scene_create(nodes, nNodes, orderedPrims, nPrims, verts, nVerts, device);
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

    std::optional<HitRecord> Intersect(const Ray &ray, float tMax = std::numeric_limits<float>::infinity(),
                                       bool *needsHost = nullptr) const {
        nnbvh_ray r = wire(ray, tMax);
        nnbvh_hit h;
        if (nnbvh_intersect_closest(scene_, &r, 1, &h) != NNBVH_OK) fatal("Intersect");
        if (needsHost) *needsHost = h.instance == -1;
        if (h.instance == -1) {
            if (!needsHost) fatal("Intersect: the ray met a host-only primitive (pass needsHost and re-trace it on the CPU)");
            return {};
        }
        if (h.prim < 0) return {};
        return HitRecord{h.prim, h.t, h.b0, h.b1, h.b2, h.instance};
    }

    bool IntersectP(const Ray &ray, float tMax = std::numeric_limits<float>::infinity(),
                    bool *needsHost = nullptr) const {
        nnbvh_ray r = wire(ray, tMax);
        uint8_t occ = 0;
        if (nnbvh_intersect_any(scene_, &r, 1, &occ, nullptr, nullptr) != NNBVH_OK)
            fatal("IntersectP");
        if (needsHost) *needsHost = occ == 2;
        if (occ == 2 && !needsHost)
            fatal("IntersectP: the ray met a host-only primitive (pass needsHost and re-trace it on the CPU)");
        return occ == 1;
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

    // == IntersectShadow of one depth + IntersectClosest of the next (the render loop issues them back to back,
    //    wavefront/integrator.cpp) in ONE launch of the traversal kernel; same results as the two calls
    void IntersectClosestAndShadowQueues(int maxRays, const nnbvh_ray_soa &rayQueue, const int32_t *dSize,
                                         const uint8_t *dPrimClass, int64_t nPrimClass, void *dHits,
                                         const nnbvh_closest_queues &out, int maxShadowRays,
                                         const nnbvh_ray_soa &shadowQueue, const int32_t *dShadowSize, const float *dLd,
                                         const float *dRu, const float *dRl, const int32_t *dPixelIndex, float *dL,
                                         int64_t nPixels, void *stream, uint8_t *dOccluded = nullptr) const {
        if (nnbvh_wavefront_intersect_closest_and_shadow(scene_, maxRays, &rayQueue, dSize, dPrimClass, nPrimClass, dHits,
                                                         &out, maxShadowRays, &shadowQueue, dShadowSize, dLd, dRu, dRl,
                                                         dPixelIndex, dL, nPixels, dOccluded, stream) != NNBVH_OK)
            fatal("IntersectClosestAndShadowQueues");
    }

    // == one wavefront iteration's independent queues as ONE launch (include/nnbvh.h)
    void TraceBatchesDevice(const nnbvh_batch *batches, int nBatches, void *stream) const {
        if (nnbvh_trace_batches_device(scene_, batches, nBatches, stream) != NNBVH_OK)
            fatal("TraceBatchesDevice");
    }
    // == WavefrontAggregate::IntersectShadowTr (wavefront/integrator.h:48-49), scenes without media
    void IntersectShadowTrQueue(const nnbvh_shading_mesh *mesh, int maxRays, const nnbvh_ray_soa &shadowQueue,
                                const int32_t *dSize, const uint8_t *dPrimClass, int64_t nPrimClass,
                                const float *dLd, const float *dRu, const float *dRl,
                                const int32_t *dPixelIndex, float *dL, int64_t nPixels, void *stream,
                                uint8_t *dState = nullptr) const {
        if (nnbvh_wavefront_intersect_shadow_tr(scene_, mesh, maxRays, &shadowQueue, dSize, dPrimClass,
                                                nPrimClass, dLd, dRu, dRl, dPixelIndex, dL, nPixels, dState,
                                                stream) != NNBVH_OK)
            fatal("IntersectShadowTrQueue");
    }
    // == WavefrontAggregate::IntersectOneRandom (wavefront/integrator.h:51-52)
    void IntersectOneRandomQueue(const nnbvh_shading_mesh *mesh, int maxItems, const float *dP0,
                                 const float *dP1, const int32_t *dMaterial, const int32_t *dSize,
                                 const int32_t *dPrimMaterial, int64_t nPrimMaterial, void *dSelHits,
                                 void *dSelRays, float *dReservoirPdf, void *stream,
                                 float *dWeightSum = nullptr) const {
        if (nnbvh_wavefront_intersect_one_random(scene_, mesh, maxItems, dP0, dP1, dMaterial, dSize,
                                                 dPrimMaterial, nPrimMaterial, dSelHits, dSelRays,
                                                 dReservoirPdf, dWeightSum, stream) != NNBVH_OK)
            fatal("IntersectOneRandomQueue");
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

    nnbvh_shading_mesh *handle() const { return mesh_; }

  private:
    nnbvh_shading_mesh *mesh_ = nullptr;
};

// KdTreeAggregate (cpu/aggregates.h:75-105): Create's parameters (aggregates.cpp:1152-1161) and the
// Primitive-shaped methods; batches as for HipBVHAggregate.
class HipKdTreeAggregate {
  public:
    HipKdTreeAggregate(const std::vector<nnbvh_prim> &prims, const std::vector<float> &verts,
                       int isectCost = 5, int traversalCost = 1, float emptyBonus = 0.5f, int maxPrims = 1,
                       int maxDepth = -1, int device = 0, const std::vector<float> *primBounds = nullptr,
                       const std::vector<float> *normals = nullptr, const std::vector<float> *uvs = nullptr,
                       const std::vector<float> *primAlpha = nullptr) {
        nnbvh_kd_build *b = nnbvh_kd_build_create(prims.data(), (int)prims.size(), verts.data(),
                                                  (int)(verts.size() / 3), primBounds ? primBounds->data() : nullptr,
                                                  isectCost, traversalCost, emptyBonus, maxPrims, maxDepth);
        if (!b) {
            HipBVHAggregate::fatal("HipKdTreeAggregate: build");
            return;
        }
        int nNodes = 0, nIdx = 0;
        const nnbvh_kd_node *nodes = nnbvh_kd_build_nodes(b, &nNodes);
        const int32_t *idx = nnbvh_kd_build_prim_indices(b, &nIdx);
        nnbvh_kd_build_bounds(b, bounds_);
        // (kd primitives stay in the caller's order: the attribute arrays pass through as they are)
        scene_ = nnbvh_kd_scene_create_with_attributes(nodes, nNodes, idx, nIdx, prims.data(), (int)prims.size(),
                                                       verts.data(), (int)(verts.size() / 3), bounds_,
                                                       normals ? normals->data() : nullptr, uvs ? uvs->data() : nullptr,
                                                       primAlpha ? primAlpha->data() : nullptr, device);
        nnbvh_kd_build_destroy(b);
        if (!scene_) HipBVHAggregate::fatal("HipKdTreeAggregate: scene_create");
    }
    // from a tree pbrt itself built: KdTreeAggregate::nodes, primitiveIndices, primitives, bounds
    HipKdTreeAggregate(const nnbvh_kd_node *nodes, int nNodes, const int32_t *primIndices, int nIndices,
                       const nnbvh_prim *prims, int nPrims, const float *verts, int nVerts,
                       const float boundsMinMax[6], int device = 0, const float *normals = nullptr,
                       const float *uvs = nullptr, const float *primAlpha = nullptr) {
        for (int k = 0; k < 6; ++k) bounds_[k] = boundsMinMax[k];
        scene_ = nnbvh_kd_scene_create_with_attributes(nodes, nNodes, primIndices, nIndices, prims, nPrims, verts, nVerts,
                                                       boundsMinMax, normals, uvs, primAlpha, device);
        if (!scene_) HipBVHAggregate::fatal("HipKdTreeAggregate: scene_create");
    }
    HipKdTreeAggregate(const HipKdTreeAggregate &) = delete;
    HipKdTreeAggregate &operator=(const HipKdTreeAggregate &) = delete;
    ~HipKdTreeAggregate() { nnbvh_kd_scene_destroy(scene_); }

    Bounds3f Bounds() const { return {{bounds_[0], bounds_[1], bounds_[2]}, {bounds_[3], bounds_[4], bounds_[5]}}; }
    std::optional<HitRecord> Intersect(const Ray &ray, float tMax = std::numeric_limits<float>::infinity(),
                                       bool *needsHost = nullptr) const {
        nnbvh_ray r{{ray.o.x, ray.o.y, ray.o.z}, tMax, {ray.d.x, ray.d.y, ray.d.z}, ray.time};
        nnbvh_hit h;
        if (nnbvh_kd_intersect_closest(scene_, &r, 1, &h) != NNBVH_OK) HipBVHAggregate::fatal("kd Intersect");
        if (needsHost) *needsHost = h.instance == -1;
        if (h.instance == -1) {  // void: a host-only primitive or an alpha re-trace hit lay on the way
            if (!needsHost) HipBVHAggregate::fatal("kd Intersect: the ray met a host-only primitive (pass needsHost)");
            return {};
        }
        if (h.prim < 0) return {};
        return HitRecord{h.prim, h.t, h.b0, h.b1, h.b2, h.instance};
    }
    bool IntersectP(const Ray &ray, float tMax = std::numeric_limits<float>::infinity(), bool *needsHost = nullptr) const {
        nnbvh_ray r{{ray.o.x, ray.o.y, ray.o.z}, tMax, {ray.d.x, ray.d.y, ray.d.z}, ray.time};
        uint8_t occ = 0;
        if (nnbvh_kd_intersect_any(scene_, &r, 1, &occ, nullptr, nullptr) != NNBVH_OK)
            HipBVHAggregate::fatal("kd IntersectP");
        if (needsHost) *needsHost = occ == 2;
        if (occ == 2 && !needsHost) HipBVHAggregate::fatal("kd IntersectP: the ray met a host-only primitive (pass needsHost)");
        return occ == 1;
    }
    void IntersectClosest(const nnbvh_ray *rays, int64_t n, nnbvh_hit *hits) const {
        if (nnbvh_kd_intersect_closest(scene_, rays, n, hits) != NNBVH_OK) HipBVHAggregate::fatal("kd IntersectClosest");
    }
    void IntersectShadow(const nnbvh_ray *rays, int64_t n, uint8_t *occluded, int32_t *nodesVisited = nullptr,
                         int32_t *primTests = nullptr) const {
        if (nnbvh_kd_intersect_any(scene_, rays, n, occluded, nodesVisited, primTests) != NNBVH_OK)
            HipBVHAggregate::fatal("kd IntersectShadow");
    }
    void IntersectClosestDevice(const void *dRays, int64_t n, void *dHits, void *stream) const {
        if (nnbvh_kd_intersect_closest_device(scene_, dRays, n, dHits, stream) != NNBVH_OK)
            HipBVHAggregate::fatal("kd IntersectClosestDevice");
    }
    void IntersectShadowDevice(const void *dRays, int64_t n, void *dOccluded, void *stream,
                               void *dNodesVisited = nullptr, void *dPrimTests = nullptr) const {
        if (nnbvh_kd_intersect_any_device(scene_, dRays, n, dOccluded, dNodesVisited, dPrimTests, stream) != NNBVH_OK)
            HipBVHAggregate::fatal("kd IntersectShadowDevice");
    }

  private:
    nnbvh_kd_scene *scene_ = nullptr;
    float bounds_[6] = {0, 0, 0, 0, 0, 0};
};

// RGBFilm's pixel accumulators (film.h:232-316): AddSample as UpdateFilm calls it, read-back, and the
// pack / unpack halves of a tile all-gather.
class HipFilm {
  public:
    HipFilm(int x0, int y0, int x1, int y1, float maxComponentValue = std::numeric_limits<float>::infinity(),
            int device = 0)
        : film_(nnbvh_film_create(x0, y0, x1, y1, maxComponentValue, device)), nPixels_((int64_t)(x1 - x0) * (y1 - y0)) {
        if (!film_) HipBVHAggregate::fatal("HipFilm");
    }
    ~HipFilm() { nnbvh_film_destroy(film_); }
    HipFilm(const HipFilm &) = delete;
    HipFilm &operator=(const HipFilm &) = delete;

    void AddSamplesDevice(const int32_t *dPx, const int32_t *dPy, const float *dRgb, int rgbStride,
                          const float *dWeight, int nPerPass, int nPasses, const int32_t *dSize, void *stream) {
        if (nnbvh_film_add_samples_device(film_, dPx, dPy, dRgb, rgbStride, dWeight, nPerPass, nPasses, dSize,
                                          stream) != NNBVH_OK)
            HipBVHAggregate::fatal("HipFilm::AddSamplesDevice");
    }
    std::vector<double> Read() {  // 4 doubles per pixel: rgbSum[3], weightSum
        std::vector<double> out((size_t)nPixels_ * 4);
        if (nnbvh_film_read(film_, out.data()) != NNBVH_OK) HipBVHAggregate::fatal("HipFilm::Read");
        return out;
    }
    void PackPixelsDevice(const int32_t *dIndex, int64_t n, void *dOut, void *stream) {
        if (nnbvh_film_pack_pixels_device(film_, dIndex, n, dOut, stream) != NNBVH_OK)
            HipBVHAggregate::fatal("HipFilm::PackPixelsDevice");
    }
    void UnpackPixelsDevice(const int32_t *dIndex, int64_t n, const void *dIn, void *stream) {
        if (nnbvh_film_unpack_pixels_device(film_, dIndex, n, dIn, stream) != NNBVH_OK)
            HipBVHAggregate::fatal("HipFilm::UnpackPixelsDevice");
    }
    nnbvh_film *handle() const { return film_; }

  private:
    nnbvh_film *film_ = nullptr;
    int64_t nPixels_ = 0;
};

// ---- per-ray calls from many threads, traced together ------------------------------------------------
// pbrt's CPU integrators call Primitive::Intersect / IntersectP once per ray from every thread of ParallelFor2D
// (cpu/integrators.cpp:152-216, 296-313).  CoalescingAggregate keeps that call shape — Intersect(ray, tMax)
// blocks and returns this ray's result — and turns the rays that are in flight at the same moment into one
// batch: a caller appends its ray to the open batch and waits; the caller that fills the batch (maxBatch) or the
// first one whose wait exceeds maxWait becomes the leader, closes the batch, traces it with one
// IntersectClosest / IntersectShadow call and wakes the others.  Rays arriving while the leader is on the GPU
// open the next batch.  Results are exactly the batched calls' (rays are independent).  A: HipBVHAggregate or
// HipKdTreeAggregate.
template <typename A>
class CoalescingAggregate {
  public:
    explicit CoalescingAggregate(const A &aggregate, size_t maxBatch = 4096,
                                 std::chrono::microseconds maxWait = std::chrono::microseconds(50))
        : agg_(aggregate), maxBatch_(maxBatch), maxWait_(maxWait) {}

    Bounds3f Bounds() const { return agg_.Bounds(); }

    std::optional<HitRecord> Intersect(const Ray &ray, float tMax = std::numeric_limits<float>::infinity(),
                                       bool *needsHost = nullptr) {
        const nnbvh_hit h = submit(closest_, wire(ray, tMax)).hit;
        if (needsHost) *needsHost = h.instance == -1;
        if (h.instance == -1) {
            if (!needsHost) HipBVHAggregate::fatal("CoalescingAggregate::Intersect: the ray met a host-only primitive (pass needsHost)");
            return {};
        }
        if (h.prim < 0) return {};
        return HitRecord{h.prim, h.t, h.b0, h.b1, h.b2, h.instance};
    }
    bool IntersectP(const Ray &ray, float tMax = std::numeric_limits<float>::infinity(), bool *needsHost = nullptr) {
        const uint8_t occ = submit(shadow_, wire(ray, tMax)).occluded;
        if (needsHost) *needsHost = occ == 2;
        if (occ == 2 && !needsHost) HipBVHAggregate::fatal("CoalescingAggregate::IntersectP: the ray met a host-only primitive (pass needsHost)");
        return occ == 1;
    }
    // batches traced so far and the rays in them (closest + shadow)
    void Stats(uint64_t *batches, uint64_t *rays) const {
        std::lock_guard<std::mutex> lock(closest_.m);
        std::lock_guard<std::mutex> lock2(shadow_.m);
        *batches = closest_.batches + shadow_.batches;
        *rays = closest_.rays + shadow_.rays;
    }

  private:
    struct Result {
        nnbvh_hit hit;
        uint8_t occluded;
    };
    struct Batch {
        std::vector<nnbvh_ray> rays;
        std::vector<nnbvh_hit> hits;
        std::vector<uint8_t> occluded;
        bool closed = false, done = false;
    };
    struct Lane {  // one kind of query (closest hit / any hit)
        bool shadow;
        mutable std::mutex m;
        std::condition_variable cv;
        std::shared_ptr<Batch> open;
        uint64_t batches = 0, rays = 0;
    };
    static nnbvh_ray wire(const Ray &ray, float tMax) {
        return nnbvh_ray{{ray.o.x, ray.o.y, ray.o.z}, tMax, {ray.d.x, ray.d.y, ray.d.z}, ray.time};
    }
    Result submit(Lane &L, const nnbvh_ray &r) {
        std::unique_lock<std::mutex> lock(L.m);
        if (!L.open) L.open = std::make_shared<Batch>();
        std::shared_ptr<Batch> b = L.open;
        const size_t index = b->rays.size();
        b->rays.push_back(r);
        bool lead = b->rays.size() >= maxBatch_;
        if (!lead) {
            // wait to be traced by someone else; the first waiter to time out leads
            const auto deadline = std::chrono::steady_clock::now() + maxWait_;
            while (!b->done && !b->closed) {
                if (L.cv.wait_until(lock, deadline) == std::cv_status::timeout && !b->closed && !b->done) {
                    lead = true;
                    break;
                }
            }
            while (!lead && !b->done) L.cv.wait(lock);
        }
        if (lead) {
            b->closed = true;
            if (L.open == b) L.open.reset();  // later rays open the next batch
            lock.unlock();
            const int64_t n = (int64_t)b->rays.size();
            if (L.shadow) {
                b->occluded.resize((size_t)n);
                agg_.IntersectShadow(b->rays.data(), n, b->occluded.data());
            } else {
                b->hits.resize((size_t)n);
                agg_.IntersectClosest(b->rays.data(), n, b->hits.data());
            }
            lock.lock();
            b->done = true;
            L.batches += 1;
            L.rays += (uint64_t)n;
            L.cv.notify_all();
        }
        Result out{};
        if (L.shadow) out.occluded = b->occluded[index];
        else out.hit = b->hits[index];
        return out;
    }

    const A &agg_;
    size_t maxBatch_;
    std::chrono::microseconds maxWait_;
    Lane closest_{false}, shadow_{true};
};

}  // namespace nnbvh
