// kd_build_gpu.hip — pbrt's KdTreeAggregate construction (cpu/aggregates.cpp:798-971) on the device.
//
// The reference builds the tree by a depth-first recursion; every node sorts the 2n bound edges of its
// primitives on one axis, sweeps them for the cheapest SAH split and hands the primitives below / above
// the split to its children.  Nothing in a node's work depends on its siblings, so the device builds the
// tree LEVEL BY LEVEL — every node of a level at once, a node = a segment of one primitive-reference
// array:
//   edges        two (key, source) pairs per reference; key = segment << 33 | orderedBits(t) << 1 | type,
//                ONE stable radix sort (rocPRIM) sorts every segment's edges by (t, type) at once
//   sweep        nBelow / nAbove at every edge from one exclusive scan of the Start flags; the cost of
//                every edge strictly inside the node in the reference's float expression; the first
//                minimum per segment through a 64-bit atomicMin of (orderedBits(cost) << 32 | index),
//                wave-reduced first
//   retries      segments without a valid edge go round again on the next axis (up to twice), as :938-942
//   children     Start edges before the chosen edge -> below child, End edges after it -> above child, in
//                sorted order (:954-960), compacted by two scans into the next level's reference array
//   leaves       their lists go to a pool; sub-tree sizes bottom-up and positions top-down then give every
//                node its place in the reference's depth-first array and every multi-primitive leaf its
//                offset in primitiveIndices (:837-850), and the two arrays are written in one pass each.
// Arithmetic: the reference's float expressions operation for operation (-ffp-contract=off).
//
// Order among EQUAL (t, type) edges.  The reference sorts with std::sort, whose order among equal keys is
// whatever its standard library's introsort leaves.  That order cannot change which edge wins (along a run
// of equal Start edges the cost grows with nBelow, along a run of equal End edges it falls with nAbove: the
// winner is the run's first / last position whoever stands there) nor which primitives go below / above —
// so node array, split planes (as floats: a tie between a -0 and a +0 edge may hand either zero to the node),
// leaf sizes and offsets are the same for every conforming sort; it only permutes primitives INSIDE
// multi-primitive leaves.  The device (and nnbvh_kd_build_create_stable, its
// host checker) keep equal edges in list order, i.e. std::stable_sort; nnbvh_kd_build_create keeps
// libstdc++'s std::sort.  tests/test_kd_build_gpu.py checks both statements.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kd_build_gpu.h"

namespace nnbvh {
namespace {

constexpr int kB = 256;

// grow-only device array that keeps its contents
template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n, bool keep, hipStream_t stream) {
        if (n <= cap) return true;
        size_t want = std::max(n, cap + cap / 2);
        T *q = nullptr;
        if (hipMalloc((void **)&q, std::max<size_t>(want, 1) * sizeof(T)) != hipSuccess) return false;
        if (keep && p && cap) {
            if (hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) {
                (void)hipFree(q);
                return false;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return true;
    }
    ~DBuf() {
        if (p) (void)hipFree(p);
    }
};

__device__ __forceinline__ unsigned ordered_bits(float f) {  // monotone float -> unsigned; -0 and +0 compare equal
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct KdParamsBuild {
    int isectCost, traversalCost, maxPrims;
    float emptyBonus;
};

// per segment state of the level being split
struct Seg {
    int start, count;  // its references in the level's list
    float box[6];      // node bounds: min xyz, max xyz
    int bad;           // badRefines inherited from the parent
    int state;         // 0 pending (wants a split), 1 leaf, 2 interior
    int axis;          // axis of the current / successful attempt
    int edgeStart;     // its edges in this attempt's edge array
    int bestOffset;
    float tSplit;
    int interiorRank;  // rank among the segments that became interior in this attempt
};

__global__ __launch_bounds__(kB) void k_classify(Seg *segs, int nSeg, int maxPrims, int depthLeft) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s >= nSeg) return;
    Seg &g = segs[s];
    const float dx = g.box[3] - g.box[0], dy = g.box[4] - g.box[1], dz = g.box[5] - g.box[2];
    g.axis = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);  // MaxDimension, vecmath.h:1306-1314
    g.state = (g.count <= maxPrims || depthLeft == 0) ? 1 : 0;  // aggregates.cpp:874-877
    g.bestOffset = -1;
}

__global__ __launch_bounds__(kB) void k_edge_counts(const Seg *segs, int nSeg, int *cnt) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s > nSeg) return;
    cnt[s] = (s < nSeg && segs[s].state == 0) ? 2 * segs[s].count : 0;
}
__global__ __launch_bounds__(kB) void k_set_edge_start(Seg *segs, int nSeg, const int *scan) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s < nSeg) segs[s].edgeStart = scan[s];
}

__global__ __launch_bounds__(kB) void k_make_edges(const Seg *segs, const int *list, const int *refSeg, int nRefs,
                                                  const float *pb, unsigned long long *keys, unsigned *vals,
                                                  unsigned long long *best) {
    const int r = blockIdx.x * kB + threadIdx.x;
    if (r >= nRefs) return;
    const int s = refSeg[r];
    const Seg &g = segs[s];
    if (g.state != 0) return;
    const int j = r - g.start;
    const float *b = pb + 6 * (long)list[r];
    const unsigned long long hi = (unsigned long long)s << 33;
    const long e = (long)g.edgeStart + 2 * j;
    keys[e] = hi | ((unsigned long long)ordered_bits(b[g.axis]) << 1);           // Start
    keys[e + 1] = hi | ((unsigned long long)ordered_bits(b[3 + g.axis]) << 1) | 1ull;  // End
    vals[e] = 2u * (unsigned)r;
    vals[e + 1] = 2u * (unsigned)r + 1u;
    if (j == 0) best[s] = ~0ull;
}

__global__ __launch_bounds__(kB) void k_start_flags(const unsigned *vals, long n, int *flags) {
    const long i = (long)blockIdx.x * kB + threadIdx.x;
    if (i <= n) flags[i] = (i < n && !(vals[i] & 1u)) ? 1 : 0;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off), hi = __shfl_xor((unsigned)(v >> 32), off);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// the sweep of aggregates.cpp:905-935 for every edge at once
__global__ __launch_bounds__(kB) void k_costs(const Seg *segs, const unsigned long long *keys, const unsigned *vals,
                                             const int *startsBefore, long nEdges, const int *list, const float *pb,
                                             KdParamsBuild P, unsigned long long *best) {
    const long i = (long)blockIdx.x * kB + threadIdx.x;
    unsigned long long packed = ~0ull;
    int s = -1;
    if (i < nEdges) {
        s = (int)(keys[i] >> 33);
        const Seg &g = segs[s];
        const unsigned v = vals[i];
        const int axis = g.axis;
        const bool isEnd = v & 1u;
        const float edgeT = pb[6 * (long)list[v >> 1] + (isEnd ? 3 : 0) + axis];
        const int local = (int)(i - g.edgeStart);
        const int nBelow = startsBefore[i] - startsBefore[g.edgeStart];
        // End edges at indices <= i = (local + 1) - Starts at indices <= i
        const int nAbove = g.count - ((local + 1) - (nBelow + (isEnd ? 0 : 1)));
        const float mn = g.box[axis], mx = g.box[3 + axis];
        if (edgeT > mn && edgeT < mx) {
            const float d[3] = {g.box[3] - g.box[0], g.box[4] - g.box[1], g.box[5] - g.box[2]};
            const float invTotalSA = 1 / (2 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]));
            const int a0 = (axis + 1) % 3, a1 = (axis + 2) % 3;
            const float belowSA = 2 * (d[a0] * d[a1] + (edgeT - mn) * (d[a0] + d[a1]));
            const float aboveSA = 2 * (d[a0] * d[a1] + (mx - edgeT) * (d[a0] + d[a1]));
            const float pBelow = belowSA * invTotalSA, pAbove = aboveSA * invTotalSA;
            const float eb = (nAbove == 0 || nBelow == 0) ? P.emptyBonus : 0;
            const float cost = P.traversalCost + P.isectCost * (1 - eb) * (pBelow * nBelow + pAbove * nAbove);
            if (cost < __builtin_inff())  // `cost < bestCost` with bestCost = inf at first; a NaN never wins
                packed = ((unsigned long long)ordered_bits(cost) << 32) | (unsigned)local;
        }
    }
    // one atomic per wavefront where the whole wavefront sweeps one segment
    const int s0 = __shfl(s, 0), s63 = __shfl(s, 63);
    if (s0 == s63 && s0 >= 0) {
        const unsigned long long m = wave_min_u64(packed);
        if ((threadIdx.x & 63) == 0 && m != ~0ull) atomicMin(&best[s0], m);
    } else if (packed != ~0ull) {
        atomicMin(&best[s], packed);
    }
}

// aggregates.cpp:938-951: retry, bad refines, leaf or interior
__global__ __launch_bounds__(kB) void k_decide(Seg *segs, int nSeg, const unsigned long long *best,
                                              const unsigned *vals, const int *list, const float *pb, int attempt,
                                              KdParamsBuild P, int *interiorFlag) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s > nSeg) return;
    if (s == nSeg) {
        interiorFlag[s] = 0;
        return;
    }
    Seg &g = segs[s];
    interiorFlag[s] = 0;
    if (g.state != 0) return;
    const unsigned long long b = best[s];
    if (b == ~0ull) {  // bestAxis == -1
        if (attempt < 2) g.axis = (g.axis + 1) % 3;
        else g.state = 1;
        return;
    }
    const unsigned ob = (unsigned)(b >> 32);
    const float bestCost = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob);
    const float leafCost = (float)((size_t)P.isectCost * (size_t)g.count);
    const int bad = g.bad + (bestCost > leafCost ? 1 : 0);
    g.bad = bad;
    if ((bestCost > 4 * leafCost && g.count < 16) || bad == 3) {
        g.state = 1;
        return;
    }
    g.state = 2;
    g.bestOffset = (int)(unsigned)b;
    const unsigned v = vals[(long)g.edgeStart + g.bestOffset];
    g.tSplit = pb[6 * (long)list[v >> 1] + ((v & 1u) ? 3 : 0) + g.axis];
    interiorFlag[s] = 1;
}

// per sorted edge of a segment that just became interior: does its primitive go below / above (:954-960)
__global__ __launch_bounds__(kB) void k_child_flags(const Seg *segs, const unsigned long long *keys, const unsigned *vals,
                                                   long nEdges, int *below, int *above) {
    const long i = (long)blockIdx.x * kB + threadIdx.x;
    if (i > nEdges) return;
    int b = 0, a = 0;
    if (i < nEdges) {
        const Seg &g = segs[(int)(keys[i] >> 33)];
        if (g.state == 2 && g.bestOffset >= 0) {
            const int local = (int)(i - g.edgeStart);
            const bool isEnd = vals[i] & 1u;
            b = (!isEnd && local < g.bestOffset) ? 1 : 0;
            a = (isEnd && local > g.bestOffset) ? 1 : 0;
        }
    }
    below[i] = b;
    above[i] = a;
}

struct NodeRec {  // a build node (breadth-first id)
    int child0, child1;  // -1: leaf
    int axis;
    float split;
    int leafN, leafOff;  // leaf: primitives and where its list starts in the pool
};

// children of the segments that became interior in this attempt: segments, boxes, node links
__global__ __launch_bounds__(kB) void k_emit_children(Seg *segs, int nSeg, const int *interiorScan, int nInterior,
                                                     const int *belowScan, const int *aboveScan, int totalBelow,
                                                     int nextRefBase, int nextSegBase, Seg *next, NodeRec *nodes,
                                                     int levelBase, int nextLevelBase) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s >= nSeg) return;
    Seg &g = segs[s];
    if (g.state != 2 || g.bestOffset < 0) return;
    const int k = interiorScan[s];
    const int es = g.edgeStart, ee = es + 2 * g.count;
    const int c0 = nextSegBase + k, c1 = nextSegBase + nInterior + k;
    Seg b0, b1;
    b0.start = nextRefBase + belowScan[es];
    b0.count = belowScan[ee] - belowScan[es];
    b1.start = nextRefBase + totalBelow + aboveScan[es];
    b1.count = aboveScan[ee] - aboveScan[es];
    for (int q = 0; q < 6; ++q) b0.box[q] = b1.box[q] = g.box[q];
    b0.box[3 + g.axis] = g.tSplit;  // bounds0.pMax[bestAxis] = tSplit, :966-968
    b1.box[g.axis] = g.tSplit;
    b0.bad = b1.bad = g.bad;
    b0.state = b1.state = 0;
    b0.axis = b1.axis = 0;
    b0.edgeStart = b1.edgeStart = 0;
    b0.bestOffset = b1.bestOffset = -1;
    b0.tSplit = b1.tSplit = 0;
    b0.interiorRank = b1.interiorRank = 0;
    next[c0] = b0;
    next[c1] = b1;
    NodeRec &nd = nodes[levelBase + s];
    nd.child0 = nextLevelBase + c0;
    nd.child1 = nextLevelBase + c1;
    nd.axis = g.axis;
    nd.split = g.tSplit;
    nd.leafN = 0;
    nd.leafOff = 0;
    g.bestOffset = -2;  // emitted: later attempts of this level leave it alone
    g.interiorRank = k;
}

__global__ __launch_bounds__(kB) void k_emit_refs(const Seg *segs, const unsigned long long *keys, const unsigned *vals,
                                                 long nEdges, const int *below, const int *above, const int *belowScan,
                                                 const int *aboveScan, int totalBelow, int nInterior, int nextRefBase,
                                                 int nextSegBase, const int *list, int *nextList, int *nextRefSeg) {
    const long i = (long)blockIdx.x * kB + threadIdx.x;
    if (i >= nEdges) return;
    if (!below[i] && !above[i]) return;
    const Seg &g = segs[(int)(keys[i] >> 33)];
    const int prim = list[vals[i] >> 1];
    if (below[i]) {
        const int at = nextRefBase + belowScan[i];
        nextList[at] = prim;
        nextRefSeg[at] = nextSegBase + g.interiorRank;
    } else {
        const int at = nextRefBase + totalBelow + aboveScan[i];
        nextList[at] = prim;
        nextRefSeg[at] = nextSegBase + nInterior + g.interiorRank;
    }
}

__global__ __launch_bounds__(kB) void k_leaf_counts(const Seg *segs, int nSeg, int *cnt) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s > nSeg) return;
    cnt[s] = (s < nSeg && segs[s].state == 1) ? segs[s].count : 0;
}
__global__ __launch_bounds__(kB) void k_emit_leaves(const Seg *segs, int nSeg, const int *leafScan, int poolBase,
                                                   NodeRec *nodes, int levelBase) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s >= nSeg || segs[s].state != 1) return;
    NodeRec &nd = nodes[levelBase + s];
    nd.child0 = nd.child1 = -1;
    nd.axis = 3;
    nd.split = 0;
    nd.leafN = segs[s].count;
    nd.leafOff = poolBase + leafScan[s];
}
__global__ __launch_bounds__(kB) void k_copy_leaf_refs(const Seg *segs, const int *list, const int *refSeg, int nRefs,
                                                      const int *leafScan, int poolBase, int levelBase, int *pool,
                                                      int *poolNode) {
    const int r = blockIdx.x * kB + threadIdx.x;
    if (r >= nRefs) return;
    const int s = refSeg[r];
    if (segs[s].state != 1) return;
    const int at = poolBase + leafScan[s] + (r - segs[s].start);
    pool[at] = list[r];
    poolNode[at] = levelBase + s;
}

// ---- depth-first layout: sub-tree sizes bottom-up, positions top-down ------------------------------
__global__ __launch_bounds__(kB) void k_sizes(const NodeRec *nodes, int base, int count, int *size, int *idx) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= count) return;
    const NodeRec &nd = nodes[base + i];
    if (nd.child0 < 0) {
        size[base + i] = 1;
        idx[base + i] = nd.leafN > 1 ? nd.leafN : 0;
    } else {
        size[base + i] = 1 + size[nd.child0] + size[nd.child1];
        idx[base + i] = idx[nd.child0] + idx[nd.child1];
    }
}
__global__ __launch_bounds__(kB) void k_positions(const NodeRec *nodes, int base, int count, const int *size,
                                                 const int *idx, int *pos, int *off) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= count) return;
    const NodeRec &nd = nodes[base + i];
    if (nd.child0 < 0) return;
    const int p = pos[base + i], o = off[base + i];
    pos[nd.child0] = p + 1;  // the below child follows its parent (:963-965)
    off[nd.child0] = o;
    pos[nd.child1] = p + 1 + size[nd.child0];
    off[nd.child1] = o + idx[nd.child0];
}
__global__ __launch_bounds__(kB) void k_write_nodes(const NodeRec *nodes, int n, const int *pos, const int *off,
                                                   const int *pool, nnbvh_kd_node *out) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const NodeRec &nd = nodes[i];
    nnbvh_kd_node o;
    if (nd.child0 < 0) {  // InitLeaf, :837-850
        o.flags = 3u | ((unsigned)nd.leafN << 2);
        const int v = nd.leafN == 0 ? 0 : (nd.leafN == 1 ? pool[nd.leafOff] : off[i]);
        o.split_or_index = (unsigned)v;
    } else {              // InitInterior, :756-759
        o.flags = (unsigned)nd.axis | ((unsigned)pos[nd.child1] << 2);
        o.split_or_index = __float_as_uint(nd.split);
    }
    out[pos[i]] = o;
}
__global__ __launch_bounds__(kB) void k_write_indices(const NodeRec *nodes, const int *pool, const int *poolNode,
                                                     int nPool, const int *off, int *out) {
    const int r = blockIdx.x * kB + threadIdx.x;
    if (r >= nPool) return;
    const int node = poolNode[r];
    const NodeRec &nd = nodes[node];
    if (nd.leafN > 1) out[off[node] + (r - nd.leafOff)] = pool[r];
}

__global__ __launch_bounds__(kB) void k_iota(int *a, int n) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) a[i] = i;
}
__global__ __launch_bounds__(kB) void k_fill(int *a, int n, int v) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) a[i] = v;
}

inline int grid(long n) { return (int)((n + kB - 1) / kB); }

}  // namespace

#define KG_CHECK(expr, what)                                                               \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            *error = std::string("device kd build: ") + what + ": " + hipGetErrorString(e_); \
            return false;                                                                  \
        }                                                                                  \
    } while (0)
#define KG_ALLOC(buf, n, keep)                                        \
    do {                                                              \
        if (!(buf).ensure((size_t)(n), keep, stream)) {               \
            *error = "device kd build: hipMalloc failed";             \
            return false;                                             \
        }                                                             \
    } while (0)

bool gpu_kd_build(const float *prim_bounds, int n_prims, const float bounds[6], int isect_cost, int traversal_cost,
                  float empty_bonus, int max_prims, int max_depth, int device, KdGpuResult *out, std::string *error) {
    const auto t0 = std::chrono::steady_clock::now();
    constexpr long kMaxRefs = 1L << 29;
    if (n_prims > kMaxRefs) {
        *error = "device kd build: more than 2^29 primitives";
        return false;
    }
    KG_CHECK(hipSetDevice(device), "hipSetDevice");
    hipStream_t stream = nullptr;
    const KdParamsBuild P{isect_cost, traversal_cost, max_prims, empty_bonus};

    DBuf<float> dPB;
    DBuf<int> listA, listB, refSegA, refSegB, scanA, scanB, scanC, flagA, flagB, flagC, pool, poolNode;
    DBuf<Seg> segA, segB;
    DBuf<unsigned long long> keys, keysS, best;
    DBuf<unsigned> vals, valsS;
    DBuf<NodeRec> nodes;
    DBuf<char> tmp;
    KG_ALLOC(dPB, (size_t)6 * n_prims, false);
    KG_CHECK(hipMemcpyAsync(dPB.p, prim_bounds, (size_t)24 * n_prims, hipMemcpyHostToDevice, stream), "upload bounds");
    KG_ALLOC(listA, n_prims, false);
    KG_ALLOC(refSegA, n_prims, false);
    hipLaunchKernelGGL(k_iota, dim3(grid(n_prims)), dim3(kB), 0, stream, listA.p, n_prims);
    hipLaunchKernelGGL(k_fill, dim3(grid(n_prims)), dim3(kB), 0, stream, refSegA.p, n_prims, 0);
    KG_ALLOC(segA, 1, false);
    Seg root{};
    root.start = 0;
    root.count = n_prims;
    std::memcpy(root.box, bounds, 24);
    KG_CHECK(hipMemcpyAsync(segA.p, &root, sizeof root, hipMemcpyHostToDevice, stream), "upload root");
    KG_ALLOC(nodes, 1, false);

    std::vector<int> levelBase, levelCount;
    int nSeg = 1, nRefs = n_prims, nNodes = 1, nPool = 0;
    DBuf<int> *list = &listA, *nextList = &listB, *refSeg = &refSegA, *nextRefSeg = &refSegB;
    DBuf<Seg> *segs = &segA, *nextSegs = &segB;
    levelBase.push_back(0);
    levelCount.push_back(1);

    auto scan = [&](int *in, int *outp, size_t n) -> bool {
        size_t bytes = 0;
        if (rocprim::exclusive_scan(nullptr, bytes, in, outp, 0, n, rocprim::plus<int>(), stream) != hipSuccess) return false;
        if (!tmp.ensure(bytes, false, stream)) return false;
        return rocprim::exclusive_scan(tmp.p, bytes, in, outp, 0, n, rocprim::plus<int>(), stream) == hipSuccess;
    };
    auto read_int = [&](const int *d, int *h) -> bool {
        return hipMemcpyAsync(h, d, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
               hipStreamSynchronize(stream) == hipSuccess;
    };

    for (int level = 0; nSeg > 0; ++level) {
        const int depthLeft = max_depth - level;
        const int base = levelBase[(size_t)level];
        hipLaunchKernelGGL(k_classify, dim3(grid(nSeg)), dim3(kB), 0, stream, segs->p, nSeg, max_prims, depthLeft);
        int nextSeg = 0, nextRefs = 0;
        KG_ALLOC(scanA, (size_t)nSeg + 1, false);
        KG_ALLOC(flagA, (size_t)nSeg + 1, false);
        KG_ALLOC(best, (size_t)nSeg, false);
        for (int attempt = 0; attempt < 3 && depthLeft > 0; ++attempt) {
            // edges of the segments that (still) want a split
            hipLaunchKernelGGL(k_edge_counts, dim3(grid(nSeg + 1)), dim3(kB), 0, stream, segs->p, nSeg, flagA.p);
            if (!scan(flagA.p, scanA.p, (size_t)nSeg + 1)) {
                *error = "device kd build: scan failed";
                return false;
            }
            int nEdgesI = 0;
            if (!read_int(scanA.p + nSeg, &nEdgesI)) {
                *error = "device kd build: read-back failed";
                return false;
            }
            if (nEdgesI == 0) break;
            const long nEdges = nEdgesI;
            hipLaunchKernelGGL(k_set_edge_start, dim3(grid(nSeg)), dim3(kB), 0, stream, segs->p, nSeg, scanA.p);
            KG_ALLOC(keys, nEdges, false);
            KG_ALLOC(keysS, nEdges, false);
            KG_ALLOC(vals, nEdges, false);
            KG_ALLOC(valsS, nEdges, false);
            hipLaunchKernelGGL(k_make_edges, dim3(grid(nRefs)), dim3(kB), 0, stream, segs->p, list->p, refSeg->p, nRefs,
                               dPB.p, keys.p, vals.p, best.p);
            int segBits = 1;
            while ((1ll << segBits) < nSeg) ++segBits;
            size_t sortBytes = 0;
            KG_CHECK(rocprim::radix_sort_pairs(nullptr, sortBytes, keys.p, keysS.p, vals.p, valsS.p, (size_t)nEdges, 0u,
                                               (unsigned)(33 + segBits), stream), "sort (size query)");
            KG_ALLOC(tmp, sortBytes, false);
            KG_CHECK(rocprim::radix_sort_pairs(tmp.p, sortBytes, keys.p, keysS.p, vals.p, valsS.p, (size_t)nEdges, 0u,
                                               (unsigned)(33 + segBits), stream), "sort");
            // the sweep
            KG_ALLOC(flagB, nEdges + 1, false);
            KG_ALLOC(scanB, nEdges + 1, false);
            hipLaunchKernelGGL(k_start_flags, dim3(grid(nEdges + 1)), dim3(kB), 0, stream, valsS.p, nEdges, flagB.p);
            if (!scan(flagB.p, scanB.p, (size_t)nEdges + 1)) {
                *error = "device kd build: scan failed";
                return false;
            }
            hipLaunchKernelGGL(k_costs, dim3(grid(nEdges)), dim3(kB), 0, stream, segs->p, keysS.p, valsS.p, scanB.p, nEdges,
                               list->p, dPB.p, P, best.p);
            hipLaunchKernelGGL(k_decide, dim3(grid(nSeg + 1)), dim3(kB), 0, stream, segs->p, nSeg, best.p, valsS.p, list->p,
                               dPB.p, attempt, P, flagA.p);
            if (!scan(flagA.p, scanA.p, (size_t)nSeg + 1)) {
                *error = "device kd build: scan failed";
                return false;
            }
            int nInterior = 0;
            if (!read_int(scanA.p + nSeg, &nInterior)) {
                *error = "device kd build: read-back failed";
                return false;
            }
            if (nInterior == 0) continue;
            // children
            KG_ALLOC(flagC, nEdges + 1, false);
            KG_ALLOC(scanC, nEdges + 1, false);
            hipLaunchKernelGGL(k_child_flags, dim3(grid(nEdges + 1)), dim3(kB), 0, stream, segs->p, keysS.p, valsS.p, nEdges,
                               flagB.p, flagC.p);
            if (!scan(flagB.p, scanB.p, (size_t)nEdges + 1) || !scan(flagC.p, scanC.p, (size_t)nEdges + 1)) {
                *error = "device kd build: scan failed";
                return false;
            }
            int totalBelow = 0, totalAbove = 0;
            if (!read_int(scanB.p + nEdges, &totalBelow) || !read_int(scanC.p + nEdges, &totalAbove)) {
                *error = "device kd build: read-back failed";
                return false;
            }
            if ((long)nextRefs + totalBelow + totalAbove > kMaxRefs) {  // 2 edges per reference, int32 scans
                *error = "device kd build: more than 2^29 primitive references on one level";
                return false;
            }
            KG_ALLOC(*nextSegs, (size_t)nextSeg + 2 * (size_t)nInterior, true);
            KG_ALLOC(*nextList, (size_t)nextRefs + totalBelow + totalAbove, true);
            KG_ALLOC(*nextRefSeg, (size_t)nextRefs + totalBelow + totalAbove, true);
            KG_ALLOC(nodes, (size_t)nNodes + (size_t)nextSeg + 2 * (size_t)nInterior, true);
            hipLaunchKernelGGL(k_emit_children, dim3(grid(nSeg)), dim3(kB), 0, stream, segs->p, nSeg, scanA.p, nInterior,
                               scanB.p, scanC.p, totalBelow, nextRefs, nextSeg, nextSegs->p, nodes.p, base, nNodes);
            hipLaunchKernelGGL(k_emit_refs, dim3(grid(nEdges)), dim3(kB), 0, stream, segs->p, keysS.p, valsS.p, nEdges,
                               flagB.p, flagC.p, scanB.p, scanC.p, totalBelow, nInterior, nextRefs, nextSeg, list->p,
                               nextList->p, nextRefSeg->p);
            nextSeg += 2 * nInterior;
            nextRefs += totalBelow + totalAbove;
        }
        // whatever did not become interior is a leaf (a segment still pending after three attempts included)
        hipLaunchKernelGGL(k_leaf_counts, dim3(grid(nSeg + 1)), dim3(kB), 0, stream, segs->p, nSeg, flagA.p);
        // (k_decide turns a pending segment into a leaf on its third failed attempt; depthLeft == 0 never gets here pending)
        if (!scan(flagA.p, scanA.p, (size_t)nSeg + 1)) {
            *error = "device kd build: scan failed";
            return false;
        }
        int leafRefs = 0;
        if (!read_int(scanA.p + nSeg, &leafRefs)) {
            *error = "device kd build: read-back failed";
            return false;
        }
        KG_ALLOC(pool, (size_t)nPool + leafRefs, true);
        KG_ALLOC(poolNode, (size_t)nPool + leafRefs, true);
        hipLaunchKernelGGL(k_emit_leaves, dim3(grid(nSeg)), dim3(kB), 0, stream, segs->p, nSeg, scanA.p, nPool, nodes.p, base);
        if (leafRefs > 0)
            hipLaunchKernelGGL(k_copy_leaf_refs, dim3(grid(nRefs)), dim3(kB), 0, stream, segs->p, list->p, refSeg->p, nRefs,
                               scanA.p, nPool, base, pool.p, poolNode.p);
        nPool += leafRefs;
        KG_CHECK(hipGetLastError(), "level kernels");
        // next level
        nNodes += nextSeg;
        if (nextSeg > 0) {
            levelBase.push_back(nNodes - nextSeg);
            levelCount.push_back(nextSeg);
        }
        std::swap(segs, nextSegs);
        std::swap(list, nextList);
        std::swap(refSeg, nextRefSeg);
        nSeg = nextSeg;
        nRefs = nextRefs;
        if (nNodes >= (1 << 30)) {
            *error = "device kd build: more than 2^30 nodes";
            return false;
        }
    }

    // depth-first layout
    DBuf<int> size, idx, pos, off;
    KG_ALLOC(size, nNodes, false);
    KG_ALLOC(idx, nNodes, false);
    KG_ALLOC(pos, nNodes, false);
    KG_ALLOC(off, nNodes, false);
    for (int l = (int)levelBase.size() - 1; l >= 0; --l)
        hipLaunchKernelGGL(k_sizes, dim3(grid(levelCount[(size_t)l])), dim3(kB), 0, stream, nodes.p, levelBase[(size_t)l],
                           levelCount[(size_t)l], size.p, idx.p);
    KG_CHECK(hipMemsetAsync(pos.p, 0, sizeof(int), stream), "memset");
    KG_CHECK(hipMemsetAsync(off.p, 0, sizeof(int), stream), "memset");
    for (size_t l = 0; l < levelBase.size(); ++l)
        hipLaunchKernelGGL(k_positions, dim3(grid(levelCount[l])), dim3(kB), 0, stream, nodes.p, levelBase[l], levelCount[l],
                           size.p, idx.p, pos.p, off.p);
    int nIndices = 0;
    if (!read_int(idx.p, &nIndices)) {
        *error = "device kd build: read-back failed";
        return false;
    }
    DBuf<nnbvh_kd_node> dOut;
    DBuf<int> dIdx;
    KG_ALLOC(dOut, nNodes, false);
    KG_ALLOC(dIdx, std::max(nIndices, 1), false);
    hipLaunchKernelGGL(k_write_nodes, dim3(grid(nNodes)), dim3(kB), 0, stream, nodes.p, nNodes, pos.p, off.p, pool.p, dOut.p);
    if (nPool > 0)
        hipLaunchKernelGGL(k_write_indices, dim3(grid(nPool)), dim3(kB), 0, stream, nodes.p, pool.p, poolNode.p, nPool, off.p,
                           dIdx.p);
    KG_CHECK(hipGetLastError(), "layout kernels");
    KG_CHECK(hipStreamSynchronize(stream), "sync");
    out->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    out->nodes.resize((size_t)nNodes);
    out->prim_indices.resize((size_t)nIndices);
    KG_CHECK(hipMemcpy(out->nodes.data(), dOut.p, (size_t)nNodes * sizeof(nnbvh_kd_node), hipMemcpyDeviceToHost), "download nodes");
    if (nIndices > 0)
        KG_CHECK(hipMemcpy(out->prim_indices.data(), dIdx.p, (size_t)nIndices * sizeof(int), hipMemcpyDeviceToHost),
                 "download indices");
    out->depth = (int)levelBase.size() - 1;
    out->levels = (int)levelBase.size();
    out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

}  // namespace nnbvh
