// kd_build.cpp — host-side construction of pbrt's KdTreeAggregate, and the C ABI of all three builders.
//
// What is computed is /root/reference/src/pbrt/cpu/aggregates.cpp:798-971:
//   KdTreeAggregate ctor   :798-835   maxDepth = round(8 + 1.3 log2 n), primitive bounds
//   KdTreeNode::InitLeaf   :837-850   flags = 3 | n << 2; one index in the node, more in primitiveIndices
//   buildTree              :852-971   per node: the bound edges of the longest axis sorted by (t, type), the SAH
//                                     cost of every edge inside the node with the empty bonus, up to two retries
//                                     on the other axes, bad-refine counting, primitives below / above
// written as three small pieces (choose_split, make_leaf, grow) over per-node primitive lists.
// The order a sort leaves among EQUAL (t, type) edges only decides the order of primitives inside
// multi-primitive leaves (kd_build_gpu.hip explains why, tests/test_kd_build_gpu.py checks it):
//   nnbvh_kd_build_create          std::sort — libstdc++'s order, what a libstdc++ build of pbrt produces
//   nnbvh_kd_build_create_stable   std::stable_sort — list order: the device builder's byte-for-byte checker
//   nnbvh_kd_build_create_gpu      the device builder (kd_build_gpu.hip)
// Parity: "unpinned" — KdTreeAggregate cannot be built from the reference here (cpu/primitive.cpp needs the
// absent nanovdb header), and the reference holds no fixture for it.
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

// ---- the builder --------------------------------------------------------------------------------------
// One node = (its primitives in list order, its box, the depth it may still use, its bad-refine count).
// choose_split() is the reference's sweep over the sorted bound edges of one axis (:896-935), emit() the
// leaf / interior bookkeeping, and grow() the recursion — below child first, so that the node array comes out
// in the reference's depth-first order with the above child's index known once the below sub-tree is complete.
struct Edge {  // BoundEdge, :781-794
    float t;
    int prim;
    bool closes;  // EdgeType::End
};
// std::tie(t, type) < std::tie(t', type') with Start < End
inline bool edge_before(const Edge &a, const Edge &b) { return a.t < b.t || (!(b.t < a.t) && !a.closes && b.closes); }

struct Split {
    int axis = -1, at = -1;  // position of the chosen edge in the sorted list
    float cost = std::numeric_limits<float>::infinity();
};

struct KdBuilder {
    int isectCost, traversalCost, maxPrims;
    float emptyBonus;
    bool stableTies = false;  // equal (t, type) edges keep list order (std::stable_sort) instead of libstdc++'s std::sort order
    const std::vector<KBox> *primBox = nullptr;
    std::vector<nnbvh_kd_node> nodes;
    std::vector<int32_t> primitiveIndices;
    std::vector<Edge> edges;  // the 2n edges of the node being split, sorted
    int maxDepthReached = 0;

    void make_leaf(int node, const std::vector<int> &prims) {  // KdTreeNode::InitLeaf, :837-850
        nnbvh_kd_node &nd = nodes[(size_t)node];
        const size_t n = prims.size();
        nd.flags = 3u | (uint32_t)(n << 2);
        int32_t v = 0;
        if (n == 1) v = prims[0];
        else if (n > 1) {
            v = (int32_t)primitiveIndices.size();
            primitiveIndices.insert(primitiveIndices.end(), prims.begin(), prims.end());
        }
        std::memcpy(&nd.split_or_index, &v, 4);
    }

    void sort_edges(const std::vector<int> &prims, int axis) {
        edges.resize(2 * prims.size());
        for (size_t i = 0; i < prims.size(); ++i) {
            const KBox &b = (*primBox)[(size_t)prims[i]];
            edges[2 * i] = Edge{b.mn[axis], prims[i], false};
            edges[2 * i + 1] = Edge{b.mx[axis], prims[i], true};
        }
        if (stableTies) std::stable_sort(edges.begin(), edges.end(), edge_before);
        else std::sort(edges.begin(), edges.end(), edge_before);
    }

    // the cheapest edge strictly inside the box on `axis`, first minimum wins; leaves `edges` sorted on that axis
    void choose_split(const std::vector<int> &prims, const KBox &box, int axis, Split *best) {
        sort_edges(prims, axis);
        const float d[3] = {box.mx[0] - box.mn[0], box.mx[1] - box.mn[1], box.mx[2] - box.mn[2]};
        const float invTotalSA = 1 / (2 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]));  // 1 / SurfaceArea, vecmath.h:1294-1297
        const int u = (axis + 1) % 3, v = (axis + 2) % 3;
        int below = 0, above = (int)prims.size();
        for (size_t i = 0; i < edges.size(); ++i) {
            const Edge &e = edges[i];
            if (e.closes) --above;
            if (e.t > box.mn[axis] && e.t < box.mx[axis]) {
                const float belowSA = 2 * (d[u] * d[v] + (e.t - box.mn[axis]) * (d[u] + d[v]));
                const float aboveSA = 2 * (d[u] * d[v] + (box.mx[axis] - e.t) * (d[u] + d[v]));
                const float pBelow = belowSA * invTotalSA, pAbove = aboveSA * invTotalSA;
                const float eb = (above == 0 || below == 0) ? emptyBonus : 0;
                const float cost = traversalCost + isectCost * (1 - eb) * (pBelow * below + pAbove * above);
                if (cost < best->cost) *best = Split{axis, (int)i, cost};
            }
            if (!e.closes) ++below;
        }
    }

    void grow(std::vector<int> prims, const KBox &box, int depthLeft, int level, int badRefines) {
        const int node = (int)nodes.size();
        nodes.emplace_back();
        maxDepthReached = std::max(maxDepthReached, level);
        if ((int)prims.size() <= maxPrims || depthLeft == 0) return make_leaf(node, prims);  // :874-877
        // longest axis first, then the other two if no edge lies inside the box (:882-942)
        const float dx = box.mx[0] - box.mn[0], dy = box.mx[1] - box.mn[1], dz = box.mx[2] - box.mn[2];
        int axis = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);  // MaxDimension, vecmath.h:1306-1314
        Split best;
        for (int attempt = 0; attempt < 3 && best.axis < 0; ++attempt, axis = (axis + 1) % 3) choose_split(prims, box, axis, &best);
        const float leafCost = (float)((size_t)isectCost * prims.size());
        if (best.cost > leafCost) ++badRefines;  // :945-951
        if ((best.cost > 4 * leafCost && prims.size() < 16) || best.axis < 0 || badRefines == 3) return make_leaf(node, prims);
        // primitives that open before the chosen edge go below, those that close after it above (:954-960)
        std::vector<int> lower, upper;
        for (int i = 0; i < best.at; ++i)
            if (!edges[(size_t)i].closes) lower.push_back(edges[(size_t)i].prim);
        for (size_t i = (size_t)best.at + 1; i < edges.size(); ++i)
            if (edges[i].closes) upper.push_back(edges[i].prim);
        const float tSplit = edges[(size_t)best.at].t;
        prims = std::vector<int>();  // the parent's list is not needed below here
        KBox lowerBox = box, upperBox = box;
        lowerBox.mx[best.axis] = upperBox.mn[best.axis] = tSplit;
        grow(std::move(lower), lowerBox, depthLeft - 1, level + 1, badRefines);
        const int aboveChild = (int)nodes.size();
        nodes[(size_t)node].flags = (uint32_t)best.axis | ((uint32_t)aboveChild << 2);  // InitInterior, :756-759
        std::memcpy(&nodes[(size_t)node].split_or_index, &tSplit, 4);
        grow(std::move(upper), upperBox, depthLeft - 1, level + 1, badRefines);
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
        const int nv = nnbvh::is_triangle_kind(p.kind) ? 3 : (p.kind == NNBVH_PRIM_BILINEAR_PATCH || nnbvh::is_alpha_patch_kind(p.kind)) ? 4 : 0;  // kinds 4 / 5: alpha-tested triangles
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
    kb.primBox = &primBounds;
    std::vector<int> all((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) all[(size_t)i] = i;
    kb.grow(std::move(all), bounds, max_depth, 0, 0);
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
