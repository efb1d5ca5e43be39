// bvh_build.cpp — host-side BVH construction for the nnbvh C ABI (no GPU code).
//
// Restates, for the flattened nnbvh_prim/vertex representation, what the reference's
// BVHAggregate constructor does (/root/reference/src/pbrt/cpu/aggregates.cpp):
//   primitive bounds + centroid        :84-93  (BVHPrimitive), shapes.cpp:294-301 (Triangle::Bounds),
//                                       shapes.cpp:1073-1081 (BilinearPatch::Bounds)
//   recursive SAH / Middle / EqualCounts :192-387 (buildRecursive; 12 buckets, leaf cost = n,
//                                       split cost = 1/2 + sum/SA; leaves on zero surface
//                                       area, single primitive or coincident centroids)
//   DFS flattening to LinearBVHNode[]  :505-522 (flattenBVH; first child = index + 1)
// The build is sequential, so leaf offsets are assigned in DFS order (the reference's
// fetch_add order when no sub-tree is built in parallel, :209, :359-370); topology is
// what the reference produces because the same libstdc++ std::partition / std::nth_element
// are driven by the same predicates on the same float32 values.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../include/nnbvh.h"
#include "nnbvh_internal.h"

namespace {

struct Box {
    float mn[3], mx[3];
    Box() {
        // Bounds3(): pMin = max float, pMax = lowest float (util/vecmath.h:1259-1264)
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::numeric_limits<float>::max();
            mx[k] = std::numeric_limits<float>::lowest();
        }
    }
    void add(const float *p) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], p[k]);
            mx[k] = std::max(mx[k], p[k]);
        }
    }
    void add(const Box &b) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], b.mn[k]);
            mx[k] = std::max(mx[k], b.mx[k]);
        }
    }
    float surface_area() const {  // util/vecmath.h:1293-1296
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    int max_dimension() const {  // util/vecmath.h:1305-1313
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx > dy && dx > dz) return 0;
        else if (dy > dz) return 1;
        else return 2;
    }
    float offset(const float *p, int dim) const {  // util/vecmath.h:1322-1331
        float o = p[dim] - mn[dim];
        if (mx[dim] > mn[dim]) o /= mx[dim] - mn[dim];
        return o;
    }
};

struct BuildPrim {
    size_t index;
    Box bounds;
    float centroid(int dim) const { return .5f * bounds.mn[dim] + .5f * bounds.mx[dim]; }
};

struct BuildNode {
    Box bounds;
    BuildNode *child[2] = {nullptr, nullptr};
    int axis = 0, first = 0, n = 0;
};

struct Builder {
    const nnbvh_prim *prims;
    int max_prims;
    int method;
    std::vector<nnbvh_prim> ordered;
    int ordered_off = 0;
    int total_nodes = 0;
    std::vector<std::unique_ptr<BuildNode[]>> pools;
    size_t pool_used = 0;
    static constexpr size_t kPool = 1 << 16;

    BuildNode *alloc() {
        if (pools.empty() || pool_used == kPool) {
            pools.emplace_back(new BuildNode[kPool]);
            pool_used = 0;
        }
        return &pools.back()[pool_used++];
    }

    BuildNode *leaf(BuildNode *node, BuildPrim *bp, size_t n, const Box &bounds) {
        int first = ordered_off;
        ordered_off += (int)n;
        for (size_t i = 0; i < n; ++i) ordered[first + i] = prims[bp[i].index];
        node->first = first;
        node->n = (int)n;
        node->bounds = bounds;
        return node;
    }

    BuildNode *build(BuildPrim *bp, size_t n) {
        BuildNode *node = alloc();
        ++total_nodes;
        Box bounds;
        for (size_t i = 0; i < n; ++i) bounds.add(bp[i].bounds);
        if (bounds.surface_area() == 0 || n == 1) return leaf(node, bp, n, bounds);

        Box cb;
        for (size_t i = 0; i < n; ++i) {
            float c[3] = {bp[i].centroid(0), bp[i].centroid(1), bp[i].centroid(2)};
            cb.add(c);
        }
        int dim = cb.max_dimension();
        if (cb.mx[dim] == cb.mn[dim]) return leaf(node, bp, n, bounds);

        size_t mid = n / 2;
        auto by_centroid = [dim](const BuildPrim &a, const BuildPrim &b) {
            return a.centroid(dim) < b.centroid(dim);
        };
        bool done = false;
        if (method == NNBVH_SPLIT_MIDDLE) {
            float pmid = (cb.mn[dim] + cb.mx[dim]) / 2;
            BuildPrim *m = std::partition(
                bp, bp + n, [dim, pmid](const BuildPrim &p) { return p.centroid(dim) < pmid; });
            mid = (size_t)(m - bp);
            done = (m != bp && m != bp + n);  // else fall through to equal counts
        }
        if (!done && (method == NNBVH_SPLIT_MIDDLE || method == NNBVH_SPLIT_EQUAL_COUNTS)) {
            mid = n / 2;
            std::nth_element(bp, bp + mid, bp + n, by_centroid);
            done = true;
        }
        if (!done) {  // SAH
            if (n <= 2) {
                mid = n / 2;
                std::nth_element(bp, bp + mid, bp + n, by_centroid);
            } else {
                constexpr int nBuckets = 12;
                int count[nBuckets] = {0};
                Box bb[nBuckets];
                auto bucket_of = [&cb, dim](const BuildPrim &p) {
                    float c[3] = {0, 0, 0};
                    c[dim] = p.centroid(dim);
                    int b = nBuckets * cb.offset(c, dim);
                    if (b == nBuckets) b = nBuckets - 1;
                    return b;
                };
                for (size_t i = 0; i < n; ++i) {
                    int b = bucket_of(bp[i]);
                    count[b]++;
                    bb[b].add(bp[i].bounds);
                }
                constexpr int nSplits = nBuckets - 1;
                float costs[nSplits] = {};
                int below = 0;
                Box bbelow;
                for (int i = 0; i < nSplits; ++i) {
                    bbelow.add(bb[i]);
                    below += count[i];
                    costs[i] += below * bbelow.surface_area();
                }
                int above = 0;
                Box babove;
                for (int i = nSplits; i >= 1; --i) {
                    babove.add(bb[i]);
                    above += count[i];
                    costs[i - 1] += above * babove.surface_area();
                }
                int best = -1;
                float min_cost = std::numeric_limits<float>::infinity();
                for (int i = 0; i < nSplits; ++i)
                    if (costs[i] < min_cost) {
                        min_cost = costs[i];
                        best = i;
                    }
                float leaf_cost = (float)n;
                min_cost = 1.f / 2.f + min_cost / bounds.surface_area();
                if ((int)n > max_prims || min_cost < leaf_cost) {
                    BuildPrim *m = std::partition(
                        bp, bp + n, [&](const BuildPrim &p) { return bucket_of(p) <= best; });
                    mid = (size_t)(m - bp);
                } else {
                    return leaf(node, bp, n, bounds);
                }
            }
        }
        node->child[0] = build(bp, mid);
        node->child[1] = build(bp + mid, n - mid);
        node->bounds = Box();
        node->bounds.add(node->child[0]->bounds);
        node->bounds.add(node->child[1]->bounds);
        node->axis = dim;
        node->n = 0;
        return node;
    }
};

int flatten(const BuildNode *node, nnbvh_linear_node *out, int *offset, int depth, int *max_depth) {
    nnbvh_linear_node *ln = &out[*offset];
    std::memcpy(ln->pmin, node->bounds.mn, 12);
    std::memcpy(ln->pmax, node->bounds.mx, 12);
    ln->pad = 0;
    int my = (*offset)++;
    if (depth > *max_depth) *max_depth = depth;
    if (node->n > 0) {
        ln->offset = node->first;
        ln->nprims = (uint16_t)node->n;
        ln->axis = 0;
    } else {
        ln->axis = (uint8_t)node->axis;
        ln->nprims = 0;
        flatten(node->child[0], out, offset, depth + 1, max_depth);
        ln->offset = flatten(node->child[1], out, offset, depth + 1, max_depth);
    }
    return my;
}

}  // namespace

struct nnbvh_build {
    std::vector<nnbvh_linear_node> nodes;
    std::vector<nnbvh_prim> ordered;
    int depth = 0;
};

extern "C" {

nnbvh_build *nnbvh_build_create(const nnbvh_prim *prims, int n_prims, const float *verts,
                                int n_verts, int max_prims_in_node, int split_method) {
    if (!prims || !verts || n_prims <= 0 || n_verts <= 0) {
        nnbvh::set_error("nnbvh_build_create: empty primitive or vertex array");
        return nullptr;
    }
    if (split_method == NNBVH_SPLIT_HLBVH) {
        nnbvh::set_error("nnbvh_build_create: HLBVH split method is not implemented");
        return nullptr;
    }
    if (split_method != NNBVH_SPLIT_SAH && split_method != NNBVH_SPLIT_MIDDLE &&
        split_method != NNBVH_SPLIT_EQUAL_COUNTS) {
        nnbvh::set_error("nnbvh_build_create: unknown split method");
        return nullptr;
    }
    std::vector<BuildPrim> bp((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) {
        const nnbvh_prim &p = prims[i];
        int nv = p.kind == NNBVH_PRIM_TRIANGLE ? 3 : p.kind == NNBVH_PRIM_BILINEAR_PATCH ? 4 : 0;
        if (!nv) {
            nnbvh::set_error("nnbvh_build_create: unknown primitive kind");
            return nullptr;
        }
        bp[i].index = (size_t)i;
        for (int k = 0; k < nv; ++k) {
            if (p.v[k] < 0 || p.v[k] >= n_verts) {
                nnbvh::set_error("nnbvh_build_create: vertex index out of range");
                return nullptr;
            }
            bp[i].bounds.add(verts + 3 * (size_t)p.v[k]);
        }
    }
    Builder b;
    b.prims = prims;
    b.max_prims = std::min(255, max_prims_in_node);  // aggregates.cpp:142
    b.method = split_method;
    b.ordered.resize((size_t)n_prims);
    BuildNode *root = b.build(bp.data(), bp.size());
    auto *out = new nnbvh_build;
    out->nodes.resize((size_t)b.total_nodes);
    int off = 0;
    flatten(root, out->nodes.data(), &off, 0, &out->depth);
    out->ordered.swap(b.ordered);
    return out;
}

const nnbvh_linear_node *nnbvh_build_nodes(const nnbvh_build *b, int *n_nodes) {
    if (!b) return nullptr;
    if (n_nodes) *n_nodes = (int)b->nodes.size();
    return b->nodes.data();
}

const nnbvh_prim *nnbvh_build_ordered_prims(const nnbvh_build *b, int *n_prims) {
    if (!b) return nullptr;
    if (n_prims) *n_prims = (int)b->ordered.size();
    return b->ordered.data();
}

int nnbvh_build_depth(const nnbvh_build *b) { return b ? b->depth : -1; }

void nnbvh_build_destroy(nnbvh_build *b) { delete b; }

}  // extern "C"
