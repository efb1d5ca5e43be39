// kd_build.cpp — host-side construction of pbrt's KdTreeAggregate (no GPU needed).
//
// Restates /root/reference/src/pbrt/cpu/aggregates.cpp:798-971 on flat arrays:
//   KdTreeAggregate ctor   :798-835   maxDepth = round(8 + 1.3 log2 n), primitive bounds, work arrays
//   KdTreeNode::InitLeaf   :837-850   flags = 3 | n << 2; one index in the node, more in primitiveIndices
//   buildTree              :852-971   per node: edges of the longest axis sorted by (t, type), SAH cost of
//                                     every edge inside the node with the empty bonus, up to two retries
//                                     on the other axes, bad-refine counting, prims below / above
// The element order std::sort leaves among equal (t, type) keys decides ties between equal-cost
// splits and the order of primitives in the leaves; the reference gets it from libstdc++'s
// std::sort, and so does this file (same comparator, same element type, same input order).
// Parity: "unpinned" — KdTreeAggregate cannot be built from the reference here (cpu/primitive.cpp
// needs the absent nanovdb header), and the reference holds no fixture for it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/nnbvh.h"
#include <hip/hip_runtime_api.h>

#include "kd_build_gpu.h"
#include "nnbvh_internal.h"

namespace {

struct KBox {
    float mn[3], mx[3];
    KBox() {  // Bounds3f(), util/vecmath.h:1259-1264
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::numeric_limits<float>::max();
            mx[k] = std::numeric_limits<float>::lowest();
        }
    }
};
// Min / Max of util/vecmath.h:425-441 are std::min / std::max per component (first of equals kept)
KBox box_of_points(const float *a, const float *b) {  // Bounds3(p1, p2), :1270
    KBox r;
    for (int k = 0; k < 3; ++k) {
        r.mn[k] = std::min(a[k], b[k]);
        r.mx[k] = std::max(a[k], b[k]);
    }
    return r;
}
KBox box_union_point(const KBox &b, const float *p) {  // Union(b, p), :1479-1484
    KBox r;
    for (int k = 0; k < 3; ++k) {
        r.mn[k] = std::min(b.mn[k], p[k]);
        r.mx[k] = std::max(b.mx[k], p[k]);
    }
    return r;
}
KBox box_union(const KBox &a, const KBox &b) {  // :1487-1492
    KBox r;
    for (int k = 0; k < 3; ++k) {
        r.mn[k] = std::min(a.mn[k], b.mn[k]);
        r.mx[k] = std::max(a.mx[k], b.mx[k]);
    }
    return r;
}

enum class EdgeType { Start, End };  // aggregates.cpp:778
struct BoundEdge {                   // :781-794
    float t;
    int primNum;
    EdgeType type;
};

struct KdBuilder {
    int isectCost, traversalCost, maxPrims;
    float emptyBonus;
    bool stableTies = false;  // equal (t, type) edges keep list order (std::stable_sort) instead of libstdc++'s std::sort order
    std::vector<nnbvh_kd_node> nodes;
    std::vector<int32_t> primitiveIndices;
    int maxDepthReached = 0;

    void init_leaf(int nodeNum, const int *primNums, size_t n) {  // :837-850
        nnbvh_kd_node &nd = nodes[(size_t)nodeNum];
        nd.flags = 3u | (uint32_t)(n << 2);
        int32_t v;
        if (n == 0)
            v = 0;
        else if (n == 1)
            v = primNums[0];
        else {
            v = (int32_t)primitiveIndices.size();
            for (size_t i = 0; i < n; ++i) primitiveIndices.push_back(primNums[i]);
        }
        std::memcpy(&nd.split_or_index, &v, 4);
    }

    void build(int nodeNum, const KBox &nodeBounds, const std::vector<KBox> &allPrimBounds, const int *primNums,
               size_t nPrimNums, int depth, int level, std::vector<BoundEdge> edges[3], int *prims0, int *prims1,
               int badRefines) {
        // :860-871 the node array grows by doubling in the reference; a vector does the same job
        if ((size_t)nodeNum != nodes.size()) std::abort();
        nodes.emplace_back();
        if (level > maxDepthReached) maxDepthReached = level;
        if ((int)nPrimNums <= maxPrims || depth == 0) {  // :874-877
            init_leaf(nodeNum, primNums, nPrimNums);
            return;
        }
        int bestAxis = -1, bestOffset = -1;
        float bestCost = std::numeric_limits<float>::infinity();
        const float leafCost = (float)((size_t)isectCost * nPrimNums);
        const float dx = nodeBounds.mx[0] - nodeBounds.mn[0], dy = nodeBounds.mx[1] - nodeBounds.mn[1],
                    dz = nodeBounds.mx[2] - nodeBounds.mn[2];
        const float invTotalSA = 1 / (2 * (dx * dy + dx * dz + dy * dz));  // SurfaceArea, vecmath.h:1294-1297
        int axis = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);           // MaxDimension, :1306-1314
        int retries = 0;
        const size_t nPrimitives = nPrimNums;
        for (;;) {  // retrySplit
            for (size_t i = 0; i < nPrimitives; ++i) {
                const int pn = primNums[i];
                const KBox &b = allPrimBounds[(size_t)pn];
                edges[axis][2 * i] = BoundEdge{b.mn[axis], pn, EdgeType::Start};
                edges[axis][2 * i + 1] = BoundEdge{b.mx[axis], pn, EdgeType::End};
            }
            const auto less = [](const BoundEdge &e0, const BoundEdge &e1) -> bool {
                return std::tie(e0.t, e0.type) < std::tie(e1.t, e1.type);
            };
            if (stableTies) std::stable_sort(edges[axis].begin(), edges[axis].begin() + 2 * nPrimitives, less);
            else std::sort(edges[axis].begin(), edges[axis].begin() + 2 * nPrimitives, less);
            int nBelow = 0, nAbove = (int)nPrimNums;
            for (size_t i = 0; i < 2 * nPrimNums; ++i) {
                if (edges[axis][i].type == EdgeType::End) --nAbove;
                const float edgeT = edges[axis][i].t;
                if (edgeT > nodeBounds.mn[axis] && edgeT < nodeBounds.mx[axis]) {
                    const float d[3] = {dx, dy, dz};
                    const int otherAxis0 = (axis + 1) % 3, otherAxis1 = (axis + 2) % 3;
                    const float belowSA = 2 * (d[otherAxis0] * d[otherAxis1] +
                                               (edgeT - nodeBounds.mn[axis]) * (d[otherAxis0] + d[otherAxis1]));
                    const float aboveSA = 2 * (d[otherAxis0] * d[otherAxis1] +
                                               (nodeBounds.mx[axis] - edgeT) * (d[otherAxis0] + d[otherAxis1]));
                    const float pBelow = belowSA * invTotalSA, pAbove = aboveSA * invTotalSA;
                    const float eb = (nAbove == 0 || nBelow == 0) ? emptyBonus : 0;
                    const float cost = traversalCost + isectCost * (1 - eb) * (pBelow * nBelow + pAbove * nAbove);
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestAxis = axis;
                        bestOffset = (int)i;
                    }
                }
                if (edges[axis][i].type == EdgeType::Start) ++nBelow;
            }
            if (bestAxis == -1 && retries < 2) {  // :938-942
                ++retries;
                axis = (axis + 1) % 3;
                continue;
            }
            break;
        }
        if (bestCost > leafCost) ++badRefines;  // :945-951
        if ((bestCost > 4 * leafCost && nPrimitives < 16) || bestAxis == -1 || badRefines == 3) {
            init_leaf(nodeNum, primNums, nPrimNums);
            return;
        }
        int n0 = 0, n1 = 0;  // :954-960
        for (int i = 0; i < bestOffset; ++i)
            if (edges[bestAxis][(size_t)i].type == EdgeType::Start) prims0[n0++] = edges[bestAxis][(size_t)i].primNum;
        for (size_t i = (size_t)bestOffset + 1; i < 2 * nPrimitives; ++i)
            if (edges[bestAxis][i].type == EdgeType::End) prims1[n1++] = edges[bestAxis][i].primNum;
        const float tSplit = edges[bestAxis][(size_t)bestOffset].t;  // :963-970
        KBox bounds0 = nodeBounds, bounds1 = nodeBounds;
        bounds0.mx[bestAxis] = bounds1.mn[bestAxis] = tSplit;
        build(nodeNum + 1, bounds0, allPrimBounds, prims0, (size_t)n0, depth - 1, level + 1, edges, prims0,
              prims1 + n1, badRefines);
        const int aboveChild = (int)nodes.size();
        nodes[(size_t)nodeNum].flags = (uint32_t)bestAxis | ((uint32_t)aboveChild << 2);  // InitInterior, :756-759
        std::memcpy(&nodes[(size_t)nodeNum].split_or_index, &tSplit, 4);
        build(aboveChild, bounds1, allPrimBounds, prims1, (size_t)n1, depth - 1, level + 1, edges, prims0,
              prims1 + n1, badRefines);
    }
};

int log2_int(uint64_t v) {  // util/math.h:420-437: index of the highest set bit
    int r = 0;
    while (v >>= 1) ++r;
    return r;
}

}  // namespace

struct nnbvh_kd_build {
    std::vector<nnbvh_kd_node> nodes;
    std::vector<int32_t> prim_indices;
    float bounds[6];
    int depth = 0;
    double build_ms[2] = {0, 0};  // device builder: on the device / incl. the download
};

// primitive bounds (Triangle::Bounds / BilinearPatch::Bounds / the caller's for host primitives), their union
// and the depth limit of :808-809 — shared by the three builders
static bool kd_prepare(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts, const float *prim_bounds,
                       int *max_depth, std::vector<KBox> *primBounds, KBox *bounds) {
    if (!prims || !verts || n_prims <= 0 || n_verts <= 0) {
        nnbvh::set_error("nnbvh_kd_build_create: empty primitive or vertex array");
        return false;
    }
    primBounds->resize((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) {
        const nnbvh_prim &p = prims[i];
        const int nv = nnbvh::is_triangle_kind(p.kind) ? 3 : p.kind == NNBVH_PRIM_BILINEAR_PATCH ? 4 : 0;  // kinds 4 / 5: alpha-tested triangles
        KBox b;
        if (p.kind == NNBVH_PRIM_HOST) {
            if (!prim_bounds) {
                nnbvh::set_error("nnbvh_kd_build_create: host primitives need prim_bounds");
                return false;
            }
            std::memcpy(b.mn, prim_bounds + 6 * (size_t)i, 12);
            std::memcpy(b.mx, prim_bounds + 6 * (size_t)i + 3, 12);
        } else if (!nv) {
            nnbvh::set_error("nnbvh_kd_build_create: unsupported primitive kind (triangles, alpha-tested triangles, patches, host primitives)");
            return false;
        } else {
            const float *v[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int k = 0; k < nv; ++k) {
                if (p.v[k] < 0 || p.v[k] >= n_verts) {
                    nnbvh::set_error("nnbvh_kd_build_create: vertex index out of range");
                    return false;
                }
                v[k] = verts + 3 * (size_t)p.v[k];
            }
            // Triangle::Bounds (shapes.cpp:294-300) / BilinearPatch::Bounds (shapes.cpp:1073-1080)
            b = nv == 3 ? box_union_point(box_of_points(v[0], v[1]), v[2])
                        : box_union(box_of_points(v[0], v[2]), box_of_points(v[1], v[3]));
        }
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(b.mn[k]) || !std::isfinite(b.mx[k])) {
                nnbvh::set_error("nnbvh_kd_build_create: non-finite vertex or primitive bounds");
                return false;
            }
        *bounds = box_union(*bounds, b);
        (*primBounds)[(size_t)i] = b;
    }
    if (*max_depth <= 0) *max_depth = (int)std::round(8 + 1.3f * log2_int((uint64_t)n_prims));  // :808-809
    if (*max_depth > nnbvh::kMaxStack) {
        nnbvh::set_error("nnbvh_kd_build_create: max_depth above the traversal stack (64, aggregates.cpp:982)");
        return false;
    }
    return true;
}

static nnbvh_kd_build *kd_build_host(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                     const float *prim_bounds, int isect_cost, int traversal_cost, float empty_bonus,
                                     int max_prims, int max_depth, bool stable_ties) {
    std::vector<KBox> primBounds;
    KBox bounds;
    if (!kd_prepare(prims, n_prims, verts, n_verts, prim_bounds, &max_depth, &primBounds, &bounds)) return nullptr;
    KdBuilder kb;
    kb.isectCost = isect_cost;
    kb.traversalCost = traversal_cost;
    kb.emptyBonus = empty_bonus;
    kb.maxPrims = max_prims;
    kb.stableTies = stable_ties;
    std::vector<BoundEdge> edges[3];
    for (int i = 0; i < 3; ++i) edges[i].resize(2 * (size_t)n_prims);
    std::vector<int> prims0((size_t)n_prims), prims1(((size_t)max_depth + 1) * (size_t)n_prims);
    std::vector<int> primNums((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) primNums[(size_t)i] = i;
    kb.build(0, bounds, primBounds, primNums.data(), (size_t)n_prims, max_depth, 0, edges, prims0.data(),
             prims1.data(), 0);
    auto *out = new nnbvh_kd_build;
    out->nodes.swap(kb.nodes);
    out->prim_indices.swap(kb.primitiveIndices);
    std::memcpy(out->bounds, bounds.mn, 12);
    std::memcpy(out->bounds + 3, bounds.mx, 12);
    out->depth = kb.maxDepthReached;
    return out;
}

extern "C" {

nnbvh_kd_build *nnbvh_kd_build_create(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                      const float *prim_bounds, int isect_cost, int traversal_cost,
                                      float empty_bonus, int max_prims, int max_depth) {
    return kd_build_host(prims, n_prims, verts, n_verts, prim_bounds, isect_cost, traversal_cost, empty_bonus, max_prims,
                         max_depth, false);
}

nnbvh_kd_build *nnbvh_kd_build_create_stable(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                             const float *prim_bounds, int isect_cost, int traversal_cost,
                                             float empty_bonus, int max_prims, int max_depth) {
    return kd_build_host(prims, n_prims, verts, n_verts, prim_bounds, isect_cost, traversal_cost, empty_bonus, max_prims,
                         max_depth, true);
}

nnbvh_kd_build *nnbvh_kd_build_create_gpu(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                          const float *prim_bounds, int isect_cost, int traversal_cost,
                                          float empty_bonus, int max_prims, int max_depth, int device) {
    std::vector<KBox> primBounds;
    KBox bounds;
    if (!kd_prepare(prims, n_prims, verts, n_verts, prim_bounds, &max_depth, &primBounds, &bounds)) return nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        nnbvh::set_error("nnbvh_kd_build_create_gpu: no usable HIP device (the device builder has no CPU fallback)");
        return nullptr;
    }
    static_assert(sizeof(KBox) == 24, "KBox is six floats");
    nnbvh::KdGpuResult r;
    std::string err;
    float b6[6];
    std::memcpy(b6, bounds.mn, 12);
    std::memcpy(b6 + 3, bounds.mx, 12);
    if (!nnbvh::gpu_kd_build(reinterpret_cast<const float *>(primBounds.data()), n_prims, b6, isect_cost, traversal_cost,
                             empty_bonus, max_prims, max_depth, device, &r, &err)) {
        nnbvh::set_error(err);
        return nullptr;
    }
    auto *out = new nnbvh_kd_build;
    out->nodes.swap(r.nodes);
    out->prim_indices.swap(r.prim_indices);
    std::memcpy(out->bounds, b6, 24);
    out->depth = r.depth;
    out->build_ms[0] = r.device_ms;
    out->build_ms[1] = r.total_ms;
    return out;
}

int nnbvh_kd_build_timing(const nnbvh_kd_build *b, double out_ms[2]) {
    if (!b || !out_ms) {
        nnbvh::set_error("nnbvh_kd_build_timing: null argument");
        return NNBVH_ERR_ARG;
    }
    out_ms[0] = b->build_ms[0];
    out_ms[1] = b->build_ms[1];
    return NNBVH_OK;
}

const nnbvh_kd_node *nnbvh_kd_build_nodes(const nnbvh_kd_build *b, int *n_nodes) {
    if (!b) return nullptr;
    if (n_nodes) *n_nodes = (int)b->nodes.size();
    return b->nodes.data();
}

const int32_t *nnbvh_kd_build_prim_indices(const nnbvh_kd_build *b, int *n_indices) {
    if (!b) return nullptr;
    if (n_indices) *n_indices = (int)b->prim_indices.size();
    return b->prim_indices.data();
}

int nnbvh_kd_build_bounds(const nnbvh_kd_build *b, float out_min_max[6]) {
    if (!b || !out_min_max) {
        nnbvh::set_error("nnbvh_kd_build_bounds: null argument");
        return NNBVH_ERR_ARG;
    }
    std::memcpy(out_min_max, b->bounds, 24);
    return NNBVH_OK;
}

int nnbvh_kd_build_depth(const nnbvh_kd_build *b) { return b ? b->depth : -1; }

void nnbvh_kd_build_destroy(nnbvh_kd_build *b) { delete b; }

}  // extern "C"
