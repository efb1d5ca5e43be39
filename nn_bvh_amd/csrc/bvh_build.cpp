// bvh_build.cpp — host-side BVH construction for the nnbvh C ABI (no GPU code).
//
// Restates, for the flattened nnbvh_prim/vertex representation, what the reference's
// BVHAggregate constructor does (/root/reference/src/pbrt/cpu/aggregates.cpp):
//   primitive bounds + centroid        :84-93  (BVHPrimitive), shapes.cpp:294-301 (Triangle::Bounds),
//                                       shapes.cpp:1073-1081 (BilinearPatch::Bounds)
//   recursive SAH / Middle / EqualCounts :192-387 (buildRecursive; 12 buckets, leaf cost = n,
//                                       split cost = 1/2 + sum/SA; leaves on zero surface
//                                       area, single primitive or coincident centroids)
//   HLBVH                              :389-503 (30-bit Morton codes of bounds centroids, 6-bit
//                                       radix sort x5 :41-82, treelets on the top 12 bits,
//                                       emitLBVH), :626-723 (buildUpperSAH over the treelets),
//                                       util/math.h:99-119 (EncodeMorton3), :508-520 (FindInterval)
//   DFS flattening to LinearBVHNode[]  :505-522 (flattenBVH; first child = index + 1)
// Leaf offsets: a sub-tree's primitives are exactly its slice of the (recursively partitioned)
// build array, so a leaf's firstPrimOffset is the slice's position — the DFS order the
// reference's fetch_add gives when nothing is built in parallel (:209, :359-370) — and needs no
// shared counter.  Like the reference (:359-370) sub-trees above 128 K primitives are built
// by separate threads; the result does not depend on the thread count.  Topology is what
// the reference produces because the same libstdc++ std::partition / std::nth_element are
// driven by the same predicates on the same float32 values.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nnbvh.h"
#include "bvh_build_gpu.h"
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

// node storage of one build thread
struct Arena {
    std::vector<std::unique_ptr<BuildNode[]>> pools;
    size_t pool_used = 0;
    int count = 0;
    static constexpr size_t kPool = 1 << 14;
    BuildNode *alloc() {
        if (pools.empty() || pool_used == kPool) {
            pools.emplace_back(new BuildNode[kPool]);
            pool_used = 0;
        }
        ++count;
        return &pools.back()[pool_used++];
    }
};

struct Builder {
    const nnbvh_prim *prims;
    int max_prims;
    int method;
    std::vector<nnbvh_prim> ordered;
    const BuildPrim *bp_base = nullptr;  // start of the build array: leaf offsets are slice positions
    int ordered_off = 0;                 // HLBVH only (sequential)
    int total_nodes = 0;
    Arena main_arena;
    std::mutex mu;
    std::vector<std::unique_ptr<Arena>> thread_arenas;
    static constexpr size_t kParallelAbove = 128 * 1024;  // aggregates.cpp:359

    BuildNode *alloc() {  // HLBVH path
        return main_arena.alloc();
    }

    BuildNode *leaf(BuildNode *node, BuildPrim *bp, size_t n, const Box &bounds) {
        int first = (int)(bp - bp_base);
        for (size_t i = 0; i < n; ++i) ordered[first + i] = prims[bp[i].index];
        node->first = first;
        node->n = (int)n;
        node->bounds = bounds;
        return node;
    }

    BuildNode *build(BuildPrim *bp, size_t n) { return build(bp, n, main_arena); }

    BuildNode *build(BuildPrim *bp, size_t n, Arena &arena) {
        BuildNode *node = arena.alloc();
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
        if (n > kParallelAbove) {
            Arena *side;
            {
                std::lock_guard<std::mutex> lock(mu);
                thread_arenas.emplace_back(new Arena);
                side = thread_arenas.back().get();
            }
            BuildNode *c0 = nullptr;
            std::thread t([&, side] { c0 = build(bp, mid, *side); });
            node->child[1] = build(bp + mid, n - mid, arena);
            t.join();
            node->child[0] = c0;
        } else {
            node->child[0] = build(bp, mid, arena);
            node->child[1] = build(bp + mid, n - mid, arena);
        }
        node->bounds = Box();
        node->bounds.add(node->child[0]->bounds);
        node->bounds.add(node->child[1]->bounds);
        node->axis = dim;
        node->n = 0;
        return node;
    }
};

// ---- HLBVH (aggregates.cpp:389-503, 626-723) ----------------------------------------------
struct MortonPrim {
    int index;
    uint32_t code;
};

inline uint32_t left_shift3(uint32_t x) {  // util/math.h:99-112
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0b00000011000000000000000011111111u;
    x = (x | (x << 8)) & 0b00000011000000001111000000001111u;
    x = (x | (x << 4)) & 0b00000011000011000011000011000011u;
    x = (x | (x << 2)) & 0b00001001001001001001001001001001u;
    return x;
}
inline uint32_t encode_morton3(float x, float y, float z) {  // util/math.h:114-119
    return (left_shift3((uint32_t)z) << 2) | (left_shift3((uint32_t)y) << 1) |
           left_shift3((uint32_t)x);
}

// The reference orders the primitives with an LSD radix sort over the 30 code bits (aggregates.cpp:41-82): a
// stable sort by the code, which is all the result depends on
void sort_by_code(std::vector<MortonPrim> *v) {
    std::stable_sort(v->begin(), v->end(), [](const MortonPrim &a, const MortonPrim &b) {
        return (a.code & 0x3fffffffu) < (b.code & 0x3fffffffu);
    });
}

struct HLBuilder {
    Builder &b;  // node pool, ordered prims, counters
    const std::vector<BuildPrim> &bp;
    std::string error;

    BuildNode *emit(const MortonPrim *mp, int n, int bitIndex) {  // emitLBVH, :451-503
        if (bitIndex == -1 || n < b.max_prims) {
            BuildNode *node = b.alloc();
            Box bounds;
            int first = b.ordered_off;
            b.ordered_off += n;
            for (int i = 0; i < n; ++i) {
                b.ordered[first + i] = b.prims[mp[i].index];
                bounds.add(bp[mp[i].index].bounds);
            }
            node->first = first;
            node->n = n;
            node->bounds = bounds;
            return node;
        }
        uint32_t mask = 1u << bitIndex;
        if ((mp[0].code & mask) == (mp[n - 1].code & mask)) return emit(mp, n, bitIndex - 1);
        // the codes are sorted and agree above this bit, so the bit reads 0...0 1...1 over the range and the first
        // primitive differs from the last: the split is where it flips (the reference's FindInterval + 1, :478-482)
        const uint32_t firstBit = mp[0].code & mask;
        const int split = (int)(std::partition_point(mp, mp + n, [&](const MortonPrim &q) { return (q.code & mask) == firstBit; }) - mp);
        BuildNode *node = b.alloc();
        node->child[0] = emit(mp, split, bitIndex - 1);
        node->child[1] = emit(mp + split, n - split, bitIndex - 1);
        node->bounds = Box();
        node->bounds.add(node->child[0]->bounds);
        node->bounds.add(node->child[1]->bounds);
        node->axis = bitIndex % 3;
        node->n = 0;
        return node;
    }

    BuildNode *upper(std::vector<BuildNode *> &roots, int start, int end) {  // buildUpperSAH
        if (!error.empty()) return nullptr;
        int n = end - start;
        if (n == 1) return roots[start];
        BuildNode *node = b.alloc();
        Box bounds, cb;
        for (int i = start; i < end; ++i) bounds.add(roots[i]->bounds);
        for (int i = start; i < end; ++i) {
            const Box &rb = roots[i]->bounds;
            float c[3] = {(rb.mn[0] + rb.mx[0]) * 0.5f, (rb.mn[1] + rb.mx[1]) * 0.5f,
                          (rb.mn[2] + rb.mx[2]) * 0.5f};
            cb.add(c);
        }
        int dim = cb.max_dimension();
        if (cb.mx[dim] == cb.mn[dim]) {  // the reference CHECK_NE-aborts here (:650)
            error = "HLBVH: treelet centroids coincide (the reference aborts on this input)";
            return nullptr;
        }
        constexpr int nBuckets = 12;
        int count[nBuckets] = {0};
        Box bb[nBuckets];
        auto bucket_of = [&cb, dim](const BuildNode *nd) {
            float centroid = (nd->bounds.mn[dim] + nd->bounds.mx[dim]) * 0.5f;
            int k = nBuckets * ((centroid - cb.mn[dim]) / (cb.mx[dim] - cb.mn[dim]));
            if (k == nBuckets) k = nBuckets - 1;
            return k;
        };
        for (int i = start; i < end; ++i) {
            int k = bucket_of(roots[i]);
            count[k]++;
            bb[k].add(roots[i]->bounds);
        }
        float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            Box b0, b1;
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) {
                b0.add(bb[j]);
                c0 += count[j];
            }
            for (int j = i + 1; j < nBuckets; ++j) {
                b1.add(bb[j]);
                c1 += count[j];
            }
            cost[i] = .125f + (c0 * b0.surface_area() + c1 * b1.surface_area()) /
                                  bounds.surface_area();
        }
        float min_cost = cost[0];
        int best = 0;
        for (int i = 1; i < nBuckets - 1; ++i)
            if (cost[i] < min_cost) {
                min_cost = cost[i];
                best = i;
            }
        BuildNode **pmid = std::partition(&roots[start], &roots[end - 1] + 1,
                                          [&](const BuildNode *nd) { return bucket_of(nd) <= best; });
        int mid = (int)(pmid - &roots[0]);
        if (mid <= start || mid >= end) {  // CHECK_GT / CHECK_LT in the reference (:715-716)
            error = "HLBVH: degenerate upper-level SAH split (the reference aborts on this input)";
            return nullptr;
        }
        node->child[0] = upper(roots, start, mid);
        node->child[1] = upper(roots, mid, end);
        if (!node->child[0] || !node->child[1]) return nullptr;
        node->bounds = Box();
        node->bounds.add(node->child[0]->bounds);
        node->bounds.add(node->child[1]->bounds);
        node->axis = dim;
        node->n = 0;
        return node;
    }

    BuildNode *build() {  // buildHLBVH, :389-449
        Box cb;
        for (const BuildPrim &p : bp) {
            float c[3] = {p.centroid(0), p.centroid(1), p.centroid(2)};
            cb.add(c);
        }
        std::vector<MortonPrim> mp(bp.size());
        for (size_t i = 0; i < bp.size(); ++i) {
            constexpr int mortonScale = 1 << 10;
            float c[3] = {bp[i].centroid(0), bp[i].centroid(1), bp[i].centroid(2)};
            mp[i].index = (int)bp[i].index;
            mp[i].code = encode_morton3(cb.offset(c, 0) * mortonScale, cb.offset(c, 1) * mortonScale,
                                        cb.offset(c, 2) * mortonScale);
        }
        sort_by_code(&mp);
        std::vector<BuildNode *> roots;
        for (size_t start = 0, end = 1; end <= mp.size(); ++end) {
            const uint32_t mask = 0b00111111111111000000000000000000u;
            if (end == mp.size() || ((mp[start].code & mask) != (mp[end].code & mask))) {
                roots.push_back(emit(&mp[start], (int)(end - start), 29 - 12));
                start = end;
            }
        }
        return upper(roots, 0, (int)roots.size());
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

// flattenBVH over the upper tree only: a treelet root takes `size` consecutive slots
struct UpperFlatten {
    const int *sizes;
    nnbvh::UpperLayout *out;
    int offset = 0;
    int run(const BuildNode *node, int depth) {
        const int my = offset;
        if (node->n < 0) {  // a treelet root, marked by hlbvh_upper_layout
            const int t = -node->n - 1;
            out->base[(size_t)t] = my;
            out->depth[(size_t)t] = depth;
            offset += sizes[t];
            return my;
        }
        ++offset;
        nnbvh_linear_node ln;
        std::memcpy(ln.pmin, node->bounds.mn, 12);
        std::memcpy(ln.pmax, node->bounds.mx, 12);
        ln.pad = 0;
        ln.axis = (uint8_t)node->axis;
        ln.nprims = 0;
        run(node->child[0], depth + 1);
        ln.offset = run(node->child[1], depth + 1);
        out->upper_index.push_back(my);
        out->upper_nodes.push_back(ln);
        return my;
    }
};

}  // namespace

namespace nnbvh {

bool hlbvh_upper_layout(const float *treelet_bounds, const int *treelet_sizes, int n_treelets,
                        UpperLayout *out, std::string *error) {
    std::vector<BuildNode> treelets((size_t)n_treelets);
    std::vector<BuildNode *> roots((size_t)n_treelets);
    for (int t = 0; t < n_treelets; ++t) {
        std::memcpy(treelets[t].bounds.mn, treelet_bounds + 6 * (size_t)t, 12);
        std::memcpy(treelets[t].bounds.mx, treelet_bounds + 6 * (size_t)t + 3, 12);
        treelets[t].n = -(t + 1);  // marker read by UpperFlatten (upper() makes nodes with n = 0)
        roots[t] = &treelets[t];
    }
    Builder b;
    std::vector<BuildPrim> none;
    HLBuilder hl{b, none, {}};
    BuildNode *root = hl.upper(roots, 0, n_treelets);
    if (!root) {
        *error = "nnbvh_build_create: " + hl.error;
        return false;
    }
    out->base.assign((size_t)n_treelets, 0);
    out->depth.assign((size_t)n_treelets, 0);
    UpperFlatten f{treelet_sizes, out};
    f.run(root, 0);
    out->total_nodes = f.offset;
    return true;
}

}  // namespace nnbvh

struct nnbvh_build {
    std::vector<nnbvh_linear_node> nodes;
    std::vector<nnbvh_prim> ordered;
    int depth = 0;
    double gpu_ms[5] = {0, 0, 0, 0, 0};
    int n_treelets = 0, n_unique_codes = 0;
};

extern "C" {

nnbvh_build *nnbvh_build_create(const nnbvh_prim *prims, int n_prims, const float *verts,
                                int n_verts, int max_prims_in_node, int split_method) {
    return nnbvh_build_create_with_bounds(prims, n_prims, verts, n_verts, nullptr, max_prims_in_node,
                                          split_method);
}

nnbvh_build *nnbvh_build_create_with_bounds(const nnbvh_prim *prims, int n_prims,
                                            const float *verts, int n_verts,
                                            const float *prim_bounds, int max_prims_in_node,
                                            int split_method) {
    if (!prims || !verts || n_prims <= 0 || n_verts <= 0) {
        nnbvh::set_error("nnbvh_build_create: empty primitive or vertex array");
        return nullptr;
    }
    if (split_method != NNBVH_SPLIT_SAH && split_method != NNBVH_SPLIT_MIDDLE &&
        split_method != NNBVH_SPLIT_EQUAL_COUNTS && split_method != NNBVH_SPLIT_HLBVH) {
        nnbvh::set_error("nnbvh_build_create: unknown split method");
        return nullptr;
    }
    const bool timing = std::getenv("NNBVH_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    std::vector<BuildPrim> bp((size_t)n_prims);
    for (int i = 0; i < n_prims; ++i) {
        const nnbvh_prim &p = prims[i];
        int nv = nnbvh::is_triangle_kind(p.kind) ? 3 : (p.kind == NNBVH_PRIM_BILINEAR_PATCH || nnbvh::is_alpha_patch_kind(p.kind)) ? 4 : 0;
        bp[i].index = (size_t)i;
        if (p.kind == NNBVH_PRIM_INSTANCE || p.kind == NNBVH_PRIM_HOST) {
            // TransformedPrimitive::Bounds() = renderFromPrimitive(child bounds) / the host
            // shape's own Bounds(): supplied by the caller
            if (!prim_bounds) {
                nnbvh::set_error("nnbvh_build_create: instance / host primitives need prim_bounds");
                return nullptr;
            }
            for (int k = 0; k < 6; ++k)
                if (!std::isfinite(prim_bounds[6 * (size_t)i + k])) {
                    nnbvh::set_error("nnbvh_build_create: non-finite vertex or primitive bounds");
                    return nullptr;
                }
            bp[i].bounds.add(prim_bounds + 6 * (size_t)i);
            bp[i].bounds.add(prim_bounds + 6 * (size_t)i + 3);
            continue;
        }
        if (!nv) {
            nnbvh::set_error("nnbvh_build_create: unknown primitive kind");
            return nullptr;
        }
        for (int k = 0; k < nv; ++k) {
            if (p.v[k] < 0 || p.v[k] >= n_verts) {
                nnbvh::set_error("nnbvh_build_create: vertex index out of range");
                return nullptr;
            }
            // Inf / NaN coordinates are malformed input, not a tree: the reference's bucket index
            // int(nBuckets * centroidBounds.Offset(c)) (aggregates.cpp:254-258) is undefined for them
            const float *v = verts + 3 * (size_t)p.v[k];
            if (!(std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]))) {
                nnbvh::set_error("nnbvh_build_create: non-finite vertex or primitive bounds");
                return nullptr;
            }
            bp[i].bounds.add(v);
        }
    }
    Builder b;
    b.prims = prims;
    b.max_prims = std::min(255, max_prims_in_node);  // aggregates.cpp:142
    b.method = split_method;
    b.ordered.resize((size_t)n_prims);
    auto t1 = now();
    BuildNode *root;
    if (split_method == NNBVH_SPLIT_HLBVH) {
        HLBuilder hl{b, bp, {}};
        root = hl.build();
        if (!root) {
            nnbvh::set_error("nnbvh_build_create: " + hl.error);
            return nullptr;
        }
    } else {
        b.bp_base = bp.data();
        root = b.build(bp.data(), bp.size());
    }
    auto t2 = now();
    auto *out = new nnbvh_build;
    b.total_nodes = b.main_arena.count;
    for (auto &a : b.thread_arenas) b.total_nodes += a->count;
    out->nodes.resize((size_t)b.total_nodes);
    int off = 0;
    flatten(root, out->nodes.data(), &off, 0, &out->depth);
    out->ordered.swap(b.ordered);
    if (timing) {
        auto ms = [](auto a, auto c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        std::fprintf(stderr, "nnbvh_build: bounds %.0f ms, tree %.0f ms, flatten %.0f ms\n", ms(t0, t1),
                     ms(t1, t2), ms(t2, now()));
    }
    return out;
}

nnbvh_build *nnbvh_build_create_gpu(const nnbvh_prim *prims, int n_prims, const float *verts,
                                    int n_verts, const float *prim_bounds, int max_prims_in_node,
                                    int split_method, int device) {
    if (!prims || !verts || n_prims <= 0 || n_verts <= 0) {
        nnbvh::set_error("nnbvh_build_create_gpu: empty primitive or vertex array");
        return nullptr;
    }
    if (split_method != NNBVH_SPLIT_SAH && split_method != NNBVH_SPLIT_HLBVH) {
        nnbvh::set_error("nnbvh_build_create_gpu: only the sah and hlbvh split methods are built on the device");
        return nullptr;
    }
    nnbvh::GpuBuildResult r;
    std::string err;
    const bool ok = split_method == NNBVH_SPLIT_SAH
                        ? nnbvh::gpu_sah(prims, n_prims, verts, n_verts, prim_bounds, max_prims_in_node, device, &r, &err)
                        : nnbvh::gpu_hlbvh(prims, n_prims, verts, n_verts, prim_bounds, max_prims_in_node, device, &r, &err);
    if (!ok) {
        nnbvh::set_error(err);
        return nullptr;
    }
    auto *out = new nnbvh_build;
    out->nodes.swap(r.nodes);
    out->ordered.swap(r.ordered);
    out->depth = r.depth;
    std::memcpy(out->gpu_ms, r.ms, sizeof r.ms);
    out->n_treelets = r.n_treelets;
    out->n_unique_codes = r.n_unique_codes;
    if (std::getenv("NNBVH_BUILD_TIMING"))
        std::fprintf(stderr, split_method == NNBVH_SPLIT_SAH
                                 ? "nnbvh_build (gpu sah): upload %.1f ms, big nodes %.1f ms, subtrees %.1f ms, layout+bounds "
                                   "%.1f ms, download %.1f ms; %d breadth-first nodes, %d subtrees\n"
                                 : "nnbvh_build (gpu hlbvh): upload %.1f ms, device sort+tree %.1f ms, upper SAH (host) "
                                   "%.1f ms, emit %.1f ms, download %.1f ms; %d distinct codes, %d treelets\n",
                     r.ms[0], r.ms[1], r.ms[2], r.ms[3], r.ms[4], r.n_unique_codes, r.n_treelets);
    return out;
}

int nnbvh_build_gpu_timing(const nnbvh_build *b, double out_ms[5]) {
    if (!b || !out_ms) return NNBVH_ERR_ARG;
    std::memcpy(out_ms, b->gpu_ms, sizeof b->gpu_ms);
    return NNBVH_OK;
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
