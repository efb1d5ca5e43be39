// bvh_build_gpu.hip — HLBVH construction on the MI355X, producing the SAME flattened tree as the
// host builder's NNBVH_SPLIT_HLBVH (bvh_build.cpp), i.e. the reference's buildHLBVH
// (/root/reference/src/pbrt/cpu/aggregates.cpp:389-503) with treelets emitted in Morton order.
//
// The reference recursion (emitLBVH, :451-503) is replaced by its closed form.  Inside a treelet
// (primitives sharing the top 12 Morton bits) emitLBVH splits a sorted range at the highest bit in
// which its codes differ, stops at ranges of fewer than maxPrimsInNode primitives, and makes one
// leaf of any run of identical codes.  That is the binary radix tree over the DISTINCT codes
// (Karras 2012: every internal node's range and split found independently by binary searches on
// common-prefix lengths), cut off where a range gets smaller than maxPrimsInNode:
//
//   1  primitive bounds, centroid bounds            k_prim_bounds     (streaming, one pass)
//   2  30-bit Morton codes (aggregates.cpp:398-408)  k_morton
//   3  stable radix sort of (code, index)           rocPRIM          (the reference's own 5x6-bit
//                                                                      LSD sort is stable too)
//   4  distinct codes ("atoms") and their ranges    k_heads + scan + k_compact
//   5  radix tree over the atoms                    k_karras
//   6  which radix nodes survive as real nodes      k_classify + 2 scans (leaf / treelet ordinals)
//   7  bounds bottom-up: leaves, then interior nodes by split bit 0..17 (children always split at
//      a lower bit, so 18 dependency-free launches; no atomics, no intra-kernel fences)
//   8  treelet roots -> host: buildUpperSAH over <= 4096 boxes (aggregates.cpp:626-723, host)
//   9  every node computes its own DFS position   k_emit: preorder = 2 x (leaves to its left in
//      the treelet) + (ancestors it hangs off on the left side), found by walking <= 18 parents
//  10  leaf-ordered primitive table = the sorted order  k_gather_prims
//
// Everything is integer / min / max work on 4-32 B records: HBM-bound streaming except the two
// pointer walks (5, 9), which touch a handful of L2-resident words per node.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "bvh_build_gpu.h"

namespace nnbvh {
namespace {

constexpr int kB = 256;
constexpr int kTreeletBit = 18;  // bits >= 18 separate treelets (mask 0x3ffc0000, aggregates.cpp:423)

enum : int { kErrVertex = 1, kErrNeedBounds = 2, kErrKind = 3, kErrLeafSize = 4, kErrNonFinite = 5 };
enum : unsigned char { kRealInterior = 1, kRealLeaf = 2, kTreeletRoot = 4 };

struct Box6 {
    float mn[3], mx[3];
};

// std::min / std::max as the host builder's Box::add applies them (first of equals is kept)
__device__ __forceinline__ float min_keep(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float max_keep(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ void box_init(Box6 &b) {  // Bounds3() (util/vecmath.h:1259-1264)
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = 3.402823466e+38f;
        b.mx[k] = -3.402823466e+38f;
    }
}
__device__ __forceinline__ void box_add(Box6 &b, const float *p) {
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = min_keep(b.mn[k], p[k]);
        b.mx[k] = max_keep(b.mx[k], p[k]);
    }
}
__device__ __forceinline__ void box_add(Box6 &b, const Box6 &o) {
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = min_keep(b.mn[k], o.mn[k]);
        b.mx[k] = max_keep(b.mx[k], o.mx[k]);
    }
}

// order-preserving float <-> unsigned map, for atomicMin/Max on floats
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- 1: primitive bounds (Triangle::Bounds / BilinearPatch::Bounds = union of the vertices,
// shapes.cpp:294-300, 1073-1081) and the bounds of their centroids (aggregates.cpp:391-394) ------
__global__ __launch_bounds__(kB) void k_prim_bounds(const nnbvh_prim *__restrict__ prims,
                                                    const float *__restrict__ verts, int nVerts,
                                                    const float *__restrict__ callerBounds, int n,
                                                    Box6 *__restrict__ pb, unsigned *cbKeys,
                                                    int *err) {
    __shared__ float red[6][kB / 64];
    float cmin[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float cmax[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB) {
        const nnbvh_prim p = prims[i];
        Box6 b;
        box_init(b);
        const int nv = (p.kind == NNBVH_PRIM_TRIANGLE || (p.kind >= NNBVH_PRIM_ALPHA_TRIANGLE && p.kind <= NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED)) ? 3 : ((p.kind == NNBVH_PRIM_BILINEAR_PATCH || (p.kind >= NNBVH_PRIM_ALPHA_PATCH && p.kind <= NNBVH_PRIM_ALPHA_PATCH_UV_SMOOTH_FLIPPED)) ? 4 : 0);
        if (p.kind == NNBVH_PRIM_INSTANCE || p.kind == NNBVH_PRIM_HOST) {
            if (!callerBounds) {
                *err = kErrNeedBounds;
            } else {
                box_add(b, callerBounds + 6 * (long)i);
                box_add(b, callerBounds + 6 * (long)i + 3);
            }
        } else if (nv == 0) {
            *err = kErrKind;
        }
        for (int k = 0; k < nv; ++k) {
            const int vi = p.v[k];
            if (vi < 0 || vi >= nVerts) {
                *err = kErrVertex;
                continue;
            }
            box_add(b, verts + 3 * (long)vi);
        }
        // non-finite coordinates (Inf / NaN vertices or caller bounds) are malformed input: the
        // bucket index int(nBuckets * offset) of aggregates.cpp:254-258 is undefined for them
        for (int k = 0; k < nv; ++k) {
            const int vi = p.v[k];
            if (vi < 0 || vi >= nVerts) continue;
            const float *v = verts + 3 * (long)vi;
            if (!(__builtin_isfinite(v[0]) && __builtin_isfinite(v[1]) && __builtin_isfinite(v[2]))) *err = kErrNonFinite;
        }
        if ((p.kind == NNBVH_PRIM_INSTANCE || p.kind == NNBVH_PRIM_HOST) && callerBounds) {
            const float *c = callerBounds + 6 * (long)i;
            for (int k = 0; k < 6; ++k)
                if (!__builtin_isfinite(c[k])) *err = kErrNonFinite;
        }
        pb[i] = b;
        for (int k = 0; k < 3; ++k) {
            const float c = .5f * b.mn[k] + .5f * b.mx[k];  // BVHPrimitive::Centroid()
            cmin[k] = fminf(cmin[k], c);
            cmax[k] = fmaxf(cmax[k], c);
        }
    }
    // block reduction (values only: the sign of a zero does not reach the Morton codes)
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off >= 1; off >>= 1) {
            cmin[k] = fminf(cmin[k], __shfl_xor(cmin[k], off));
            cmax[k] = fmaxf(cmax[k], __shfl_xor(cmax[k], off));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) {
            red[k][wave] = cmin[k];
            red[3 + k][wave] = cmax[k];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < kB / 64; ++w)
            v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        if (threadIdx.x < 3) atomicMin(&cbKeys[threadIdx.x], f2key(v));
        else atomicMax(&cbKeys[threadIdx.x], f2key(v));
    }
}

// ---- 2: Morton codes (aggregates.cpp:398-408; EncodeMorton3 / LeftShift3 util/math.h:99-119;
// Bounds3::Offset util/vecmath.h:1322-1331) --------------------------------------------------------
__device__ __forceinline__ unsigned left_shift3(unsigned x) {
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0b00000011000000000000000011111111u;
    x = (x | (x << 8)) & 0b00000011000000001111000000001111u;
    x = (x | (x << 4)) & 0b00000011000011000011000011000011u;
    x = (x | (x << 2)) & 0b00001001001001001001001001001001u;
    return x;
}

__global__ __launch_bounds__(kB) void k_morton(const Box6 *__restrict__ pb,
                                               const unsigned *__restrict__ cbKeys, int n,
                                               unsigned *__restrict__ codes, int *__restrict__ idx) {
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
        mn[k] = key2f(cbKeys[k]);
        mx[k] = key2f(cbKeys[3 + k]);
    }
    for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB) {
        const Box6 b = pb[i];
        unsigned q[3];
        for (int k = 0; k < 3; ++k) {
            const float c = .5f * b.mn[k] + .5f * b.mx[k];
            float o = c - mn[k];
            if (mx[k] > mn[k]) o /= mx[k] - mn[k];
            q[k] = (unsigned)(o * 1024.0f);  // mortonScale = 1 << 10
        }
        codes[i] = (left_shift3(q[2]) << 2) | (left_shift3(q[1]) << 1) | left_shift3(q[0]);
        idx[i] = i;
    }
}

// ---- 4: runs of identical codes -------------------------------------------------------------------
__global__ __launch_bounds__(kB) void k_heads(const unsigned *__restrict__ codes, int n,
                                              int *__restrict__ head) {
    for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
        head[i] = (i == 0 || codes[i] != codes[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(kB) void k_compact(const unsigned *__restrict__ codes,
                                                const int *__restrict__ head,
                                                const int *__restrict__ excl, int n,
                                                int *__restrict__ pstart, unsigned *__restrict__ ucode,
                                                int *mOut) {
    for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB) {
        if (head[i]) {
            pstart[excl[i]] = i;
            ucode[excl[i]] = codes[i];
        }
        if (i == n - 1) {
            const int m = excl[i] + head[i];
            *mOut = m;
            pstart[m] = n;
        }
    }
}

// ---- 5: binary radix tree over the m distinct codes.  Internal node i in [0, m-1); node index
// space "v": internal nodes 0..m-2, atom a -> (m-1)+a.  Split position == the reference's
// FindInterval result (first element whose bit differs from the range's first, :478-482). --------
__global__ __launch_bounds__(kB) void k_karras(const unsigned *__restrict__ ucode, int m,
                                               int *__restrict__ firstA, int *__restrict__ lastA,
                                               int *__restrict__ leftV, int *__restrict__ rightV,
                                               unsigned char *__restrict__ bit,
                                               int *__restrict__ parentV) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= m - 1) return;
    const unsigned ci = ucode[i];
    auto delta = [&](int j) -> int { return (j < 0 || j >= m) ? -1 : __clz((int)(ci ^ ucode[j])); };
    const int d = (delta(i + 1) - delta(i - 1)) > 0 ? 1 : -1;
    const int dmin = delta(i - d);
    int lmax = 2;
    while (delta(i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lv = (lo == gamma) ? (m - 1) + gamma : gamma;
    const int rv = (hi == gamma + 1) ? (m - 1) + gamma + 1 : gamma + 1;
    firstA[i] = lo;
    lastA[i] = hi;
    leftV[i] = lv;
    rightV[i] = rv;
    bit[i] = (unsigned char)(31 - dnode);
    parentV[lv] = i;
    parentV[rv] = i;
    if (i == 0) parentV[0] = -1;
}

// ---- 6: which radix nodes are nodes of the reference's tree --------------------------------------
// internal node: interior iff it lies inside a treelet (split bit < 18) and holds >= maxPrims
// primitives (emitLBVH's leaf rule, :453).  Any node inside a treelet whose parent is such an
// interior node (or which is its treelet's root) and which is not interior itself is a leaf.
__global__ __launch_bounds__(kB) void k_classify(int m, int maxPrims, const int *__restrict__ pstart,
                                                 const int *__restrict__ firstA,
                                                 const int *__restrict__ lastA,
                                                 const unsigned char *__restrict__ bit,
                                                 const int *__restrict__ parentV,
                                                 unsigned char *__restrict__ kind,
                                                 int *__restrict__ leafHead,
                                                 int *__restrict__ treeletHead, int *err) {
    const int v = blockIdx.x * kB + threadIdx.x;
    if (v >= 2 * m - 1) return;
    const bool internal = v < m - 1;
    const int first = internal ? firstA[v] : v - (m - 1);
    const int last = internal ? lastA[v] : first;
    const int n = pstart[last + 1] - pstart[first];
    const int p = parentV[v];
    const bool inTreelet = !internal || bit[v] < kTreeletBit;
    const bool parentInTreelet = p >= 0 && bit[p] < kTreeletBit;
    const bool treeletRoot = inTreelet && !parentInTreelet;
    const bool realInterior = internal && inTreelet && n >= maxPrims;
    bool parentInterior = false;
    if (parentInTreelet) parentInterior = pstart[lastA[p] + 1] - pstart[firstA[p]] >= maxPrims;
    const bool realLeaf = inTreelet && !realInterior && (treeletRoot || parentInterior);
    kind[v] = (realInterior ? kRealInterior : 0) | (realLeaf ? kRealLeaf : 0) |
              (treeletRoot ? kTreeletRoot : 0);
    if (realLeaf) {
        leafHead[first] = 1;
        if (n > 65535) *err = kErrLeafSize;  // LinearBVHNode::nPrimitives is 16 bits (:134)
    }
    if (treeletRoot) treeletHead[first] = 1;
}

// ---- 7: bounds ---------------------------------------------------------------------------------------
constexpr int kBigLeaf = 64;  // leaves above this size are folded by a whole wavefront

__global__ __launch_bounds__(kB) void k_leaf_bounds(int m, const unsigned char *__restrict__ kind,
                                                    const int *__restrict__ pstart,
                                                    const int *__restrict__ firstA,
                                                    const int *__restrict__ lastA,
                                                    const int *__restrict__ idxSorted,
                                                    const Box6 *__restrict__ pb,
                                                    Box6 *__restrict__ nodeBox) {
    const int v = blockIdx.x * kB + threadIdx.x;
    if (v >= 2 * m - 1 || !(kind[v] & kRealLeaf)) return;
    const bool internal = v < m - 1;
    const int first = internal ? firstA[v] : v - (m - 1);
    const int last = internal ? lastA[v] : first;
    const int begin = pstart[first], end = pstart[last + 1];
    if (end - begin > kBigLeaf) return;  // k_big_leaf_bounds
    Box6 b;
    box_init(b);
    for (int j = begin; j < end; ++j) box_add(b, pb[idxSorted[j]]);  // :456-463
    nodeBox[v] = b;
}

// One wavefront per large leaf (runs of identical Morton codes can hold thousands of primitives).
// The result must equal the sequential fold's, which keeps the FIRST of equal values (+0 / -0):
// lanes fold contiguous chunks in order, and the cross-lane reduction always combines
// (lower lane = earlier elements) on the left.
__global__ __launch_bounds__(kB) void k_big_leaf_bounds(int m, const unsigned char *__restrict__ kind,
                                                        const int *__restrict__ pstart,
                                                        const int *__restrict__ firstA,
                                                        const int *__restrict__ lastA,
                                                        const int *__restrict__ idxSorted,
                                                        const Box6 *__restrict__ pb,
                                                        Box6 *__restrict__ nodeBox) {
    const int v = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);  // wave-uniform
    if (v >= 2 * m - 1 || !(kind[v] & kRealLeaf)) return;
    const bool internal = v < m - 1;
    const int first = internal ? firstA[v] : v - (m - 1);
    const int last = internal ? lastA[v] : first;
    const int begin = pstart[first], n = pstart[last + 1] - begin;
    if (n <= kBigLeaf) return;
    const int lane = threadIdx.x & 63;
    const int chunk = (n + 63) / 64;
    const int lo = lane * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    Box6 b;
    box_init(b);
    for (int j = lo; j < hi; ++j) box_add(b, pb[idxSorted[begin + j]]);
    for (int off = 1; off < 64; off <<= 1) {
        Box6 o;
        for (int k = 0; k < 3; ++k) {
            o.mn[k] = __shfl_down(b.mn[k], off);
            o.mx[k] = __shfl_down(b.mx[k], off);
        }
        if (lane + off < 64) box_add(b, o);  // b = earlier elements (kept on ties), o = later ones
    }
    if (lane == 0) nodeBox[v] = b;
}

__global__ __launch_bounds__(kB) void k_interior_bounds(int m, int level,
                                                        const unsigned char *__restrict__ kind,
                                                        const unsigned char *__restrict__ bit,
                                                        const int *__restrict__ leftV,
                                                        const int *__restrict__ rightV,
                                                        Box6 *nodeBox) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= m - 1 || !(kind[i] & kRealInterior) || bit[i] != level) return;
    Box6 b;
    box_init(b);
    box_add(b, nodeBox[leftV[i]]);  // :495-498 Union(child0, child1)
    box_add(b, nodeBox[rightV[i]]);
    nodeBox[i] = b;
}

// ---- 8: treelet table for the host's upper SAH build -------------------------------------------------
__device__ __forceinline__ int subtree_size(int v, int m, const unsigned char *kind, const int *firstA,
                                            const int *lastA, const int *leafOrd) {
    if (!(kind[v] & kRealInterior)) return 1;
    return 2 * (leafOrd[lastA[v] + 1] - leafOrd[firstA[v]]) - 1;
}

__global__ __launch_bounds__(kB) void k_treelets(int m, const unsigned char *__restrict__ kind,
                                                 const int *__restrict__ firstA,
                                                 const int *__restrict__ lastA,
                                                 const int *__restrict__ leafOrd,
                                                 const int *__restrict__ treeOrd,
                                                 const Box6 *__restrict__ nodeBox,
                                                 Box6 *__restrict__ tbox, int *__restrict__ tsize) {
    const int v = blockIdx.x * kB + threadIdx.x;
    if (v >= 2 * m - 1 || !(kind[v] & kTreeletRoot)) return;
    const int first = v < m - 1 ? firstA[v] : v - (m - 1);
    const int t = treeOrd[first];
    tbox[t] = nodeBox[v];
    tsize[t] = subtree_size(v, m, kind, firstA, lastA, leafOrd);
}

// ---- 9: flattenBVH's DFS order (aggregates.cpp:505-522) without a traversal ---------------------------
__global__ __launch_bounds__(kB) void k_emit(int m, const unsigned char *__restrict__ kind,
                                             const unsigned char *__restrict__ bit,
                                             const int *__restrict__ pstart,
                                             const int *__restrict__ firstA,
                                             const int *__restrict__ lastA,
                                             const int *__restrict__ leftV,
                                             const int *__restrict__ parentV,
                                             const int *__restrict__ leafOrd,
                                             const int *__restrict__ treeOrd,
                                             const int *__restrict__ tbase,
                                             const int *__restrict__ tdepth,
                                             const Box6 *__restrict__ nodeBox,
                                             nnbvh_linear_node *__restrict__ nodes, int *maxDepth) {
    __shared__ int blockDepth;
    if (threadIdx.x == 0) blockDepth = 0;
    __syncthreads();
    const int v = blockIdx.x * kB + threadIdx.x;
    if (v < 2 * m - 1 && (kind[v] & (kRealInterior | kRealLeaf))) {
        const bool internal = v < m - 1;
        const int first = internal ? firstA[v] : v - (m - 1);
        const int last = internal ? lastA[v] : first;
        int cur = v, leftTurns = 0, depth = 0;
        while (!(kind[cur] & kTreeletRoot)) {
            const int p = parentV[cur];
            leftTurns += (leftV[p] == cur) ? 1 : 0;
            ++depth;
            cur = p;
        }
        const int rootFirst = cur < m - 1 ? firstA[cur] : cur - (m - 1);
        const int t = treeOrd[rootFirst];
        const int flat = tbase[t] + 2 * (leafOrd[first] - leafOrd[rootFirst]) + leftTurns;
        const Box6 b = nodeBox[v];
        nnbvh_linear_node out;
        for (int k = 0; k < 3; ++k) {
            out.pmin[k] = b.mn[k];
            out.pmax[k] = b.mx[k];
        }
        out.pad = 0;
        if (kind[v] & kRealLeaf) {
            out.offset = pstart[first];  // leaves are emitted in sorted order: offset = range start
            out.nprims = (uint16_t)(pstart[last + 1] - pstart[first]);
            out.axis = 0;
            atomicMax(&blockDepth, tdepth[t] + depth);
        } else {
            out.offset = flat + 1 + subtree_size(leftV[v], m, kind, firstA, lastA, leafOrd);
            out.nprims = 0;
            out.axis = (uint8_t)(bit[v] % 3);  // :499
        }
        nodes[flat] = out;
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockDepth > 0) atomicMax(maxDepth, blockDepth);
}

// ---- 10 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kB) void k_gather_prims(const nnbvh_prim *__restrict__ prims,
                                                     const int *__restrict__ idxSorted, int n,
                                                     nnbvh_prim *__restrict__ ordered) {
    for (int i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
        ordered[i] = prims[idxSorted[i]];
}

__global__ __launch_bounds__(kB) void k_scatter_nodes(const int *__restrict__ index,
                                                      const nnbvh_linear_node *__restrict__ src, int n,
                                                      nnbvh_linear_node *__restrict__ nodes) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) nodes[index[i]] = src[i];
}

// ---------------------------------------------------------------------------------------------------------
struct DevMem {  // frees everything it handed out
    std::vector<void *> ptrs;
    std::string *error;
    bool ok = true;
    template <typename T>
    T *get(size_t count) {
        void *p = nullptr;
        if (!ok) return nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
            ok = false;
            *error = "gpu build: hipMalloc failed";
            return nullptr;
        }
        ptrs.push_back(p);
        return (T *)p;
    }
    void release(void *p) {  // ownership goes to the caller
        for (void *&q : ptrs)
            if (q == p) q = nullptr;
    }
    ~DevMem() {
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    }
};

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int grid_for(long n, int cap = 4096) {
    long b = (n + kB - 1) / kB;
    return (int)std::max<long>(1, std::min<long>(b, cap));
}
int grid_all(long n) { return (int)std::max<long>(1, (n + kB - 1) / kB); }

}  // namespace

#define GB_CHECK(expr, what)                                         \
    do {                                                             \
        hipError_t e_ = (expr);                                      \
        if (e_ != hipSuccess) {                                      \
            *error = std::string("gpu build: ") + what + ": " + hipGetErrorString(e_); \
            return false;                                            \
        }                                                            \
    } while (0)

bool gpu_hlbvh(const nnbvh_prim *prims, int n, const float *verts, int n_verts,
               const float *prim_bounds, int max_prims_in_node, int device, GpuBuildResult *out,
               std::string *error) {
    int prev = 0;
    GB_CHECK(hipGetDevice(&prev), "hipGetDevice");
    GB_CHECK(hipSetDevice(device), "hipSetDevice");
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore{prev};
    const int maxPrims = std::min(255, max_prims_in_node);  // aggregates.cpp:142
    hipStream_t stream = nullptr;
    DevMem mem;
    mem.error = error;

    auto t0 = std::chrono::steady_clock::now();
    nnbvh_prim *dPrims = mem.get<nnbvh_prim>(n);
    float *dVerts = mem.get<float>(3 * (size_t)n_verts);
    float *dCaller = prim_bounds ? mem.get<float>(6 * (size_t)n) : nullptr;
    Box6 *dPb = mem.get<Box6>(n);
    unsigned *dCodes = mem.get<unsigned>(n), *dCodesS = mem.get<unsigned>(n);
    int *dIdx = mem.get<int>(n), *dIdxS = mem.get<int>(n);
    int *dHead = mem.get<int>(n), *dExcl = mem.get<int>(n);
    int *dPstart = mem.get<int>((size_t)n + 1);
    unsigned *dUcode = mem.get<unsigned>(n);
    int *dScalars = mem.get<int>(16);  // [0..5] centroid-bound keys, 6 err, 7 m, 8 maxDepth
    if (!mem.ok) return false;
    GB_CHECK(hipMemcpyAsync(dPrims, prims, (size_t)n * sizeof(nnbvh_prim), hipMemcpyHostToDevice, stream), "copy prims");
    GB_CHECK(hipMemcpyAsync(dVerts, verts, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, stream), "copy verts");
    if (dCaller)
        GB_CHECK(hipMemcpyAsync(dCaller, prim_bounds, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream), "copy bounds");
    const int init[16] = {-1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // min keys all-ones, max keys 0
    GB_CHECK(hipMemcpyAsync(dScalars, init, sizeof init, hipMemcpyHostToDevice, stream), "init scalars");
    GB_CHECK(hipStreamSynchronize(stream), "sync after upload");
    out->ms[0] = ms_since(t0);

    t0 = std::chrono::steady_clock::now();
    unsigned *dCb = (unsigned *)dScalars;
    int *dErr = dScalars + 6, *dM = dScalars + 7, *dMaxDepth = dScalars + 8;
    hipLaunchKernelGGL(k_prim_bounds, dim3(grid_for(n, 1024)), dim3(kB), 0, stream, dPrims, dVerts, n_verts,
                       dCaller, n, dPb, dCb, dErr);
    hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(kB), 0, stream, dPb, dCb, n, dCodes, dIdx);
    size_t tmpBytes = 0, scanBytes = 0;
    GB_CHECK(rocprim::radix_sort_pairs(nullptr, tmpBytes, dCodes, dCodesS, dIdx, dIdxS, (size_t)n, 0u, 30u, stream),
             "radix sort (size query)");
    GB_CHECK(rocprim::exclusive_scan(nullptr, scanBytes, dHead, dExcl, 0, (size_t)n + 1, rocprim::plus<int>(), stream),
             "scan (size query)");
    void *dTmp = mem.get<char>(std::max(tmpBytes, scanBytes));
    if (!mem.ok) return false;
    GB_CHECK(rocprim::radix_sort_pairs(dTmp, tmpBytes, dCodes, dCodesS, dIdx, dIdxS, (size_t)n, 0u, 30u, stream),
             "radix sort");
    hipLaunchKernelGGL(k_heads, dim3(grid_for(n)), dim3(kB), 0, stream, dCodesS, n, dHead);
    GB_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dHead, dExcl, 0, (size_t)n, rocprim::plus<int>(), stream), "scan");
    hipLaunchKernelGGL(k_compact, dim3(grid_for(n)), dim3(kB), 0, stream, dCodesS, dHead, dExcl, n, dPstart, dUcode, dM);
    int scal[16];
    GB_CHECK(hipMemcpyAsync(scal, dScalars, sizeof scal, hipMemcpyDeviceToHost, stream), "read scalars");
    GB_CHECK(hipStreamSynchronize(stream), "sync after sort");
    if (scal[6] != 0) {
        *error = scal[6] == kErrVertex       ? "nnbvh_build_create: vertex index out of range"
                 : scal[6] == kErrNeedBounds ? "nnbvh_build_create: instance / host primitives need prim_bounds"
                 : scal[6] == kErrNonFinite  ? "nnbvh_build_create: non-finite vertex or primitive bounds"
                                             : "nnbvh_build_create: unknown primitive kind";
        return false;
    }
    const int m = scal[7];
    out->n_unique_codes = m;
    const int nv = 2 * m - 1;

    int *dFirst = mem.get<int>(m), *dLast = mem.get<int>(m), *dLeft = mem.get<int>(m), *dRight = mem.get<int>(m);
    unsigned char *dBit = mem.get<unsigned char>(m), *dKind = mem.get<unsigned char>(nv);
    int *dParent = mem.get<int>(nv);
    int *dLeafHead = mem.get<int>((size_t)m + 1), *dTreeHead = mem.get<int>((size_t)m + 1);
    int *dLeafOrd = mem.get<int>((size_t)m + 1), *dTreeOrd = mem.get<int>((size_t)m + 1);
    Box6 *dNodeBox = mem.get<Box6>(nv);
    Box6 *dTbox = mem.get<Box6>(4096);
    int *dTsize = mem.get<int>(4096), *dTbase = mem.get<int>(4096), *dTdepth = mem.get<int>(4096);
    if (!mem.ok) return false;
    GB_CHECK(hipMemsetAsync(dParent, 0xff, (size_t)nv * sizeof(int), stream), "memset parents");  // -1: the root
    GB_CHECK(hipMemsetAsync(dLeafHead, 0, ((size_t)m + 1) * sizeof(int), stream), "memset");
    GB_CHECK(hipMemsetAsync(dTreeHead, 0, ((size_t)m + 1) * sizeof(int), stream), "memset");
    if (m > 1)
        hipLaunchKernelGGL(k_karras, dim3(grid_all(m - 1)), dim3(kB), 0, stream, dUcode, m, dFirst, dLast, dLeft,
                           dRight, dBit, dParent);
    hipLaunchKernelGGL(k_classify, dim3(grid_all(nv)), dim3(kB), 0, stream, m, maxPrims, dPstart, dFirst, dLast, dBit,
                       dParent, dKind, dLeafHead, dTreeHead, dErr);
    GB_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dLeafHead, dLeafOrd, 0, (size_t)m + 1, rocprim::plus<int>(), stream),
             "scan leaves");
    GB_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dTreeHead, dTreeOrd, 0, (size_t)m + 1, rocprim::plus<int>(), stream),
             "scan treelets");
    hipLaunchKernelGGL(k_leaf_bounds, dim3(grid_all(nv)), dim3(kB), 0, stream, m, dKind, dPstart, dFirst, dLast, dIdxS,
                       dPb, dNodeBox);
    hipLaunchKernelGGL(k_big_leaf_bounds, dim3((nv + kB / 64 - 1) / (kB / 64)), dim3(kB), 0, stream, m, dKind, dPstart,
                       dFirst, dLast, dIdxS, dPb, dNodeBox);
    if (m > 1)
        for (int level = 0; level < kTreeletBit; ++level)
            hipLaunchKernelGGL(k_interior_bounds, dim3(grid_all(m - 1)), dim3(kB), 0, stream, m, level, dKind, dBit,
                               dLeft, dRight, dNodeBox);
    hipLaunchKernelGGL(k_treelets, dim3(grid_all(nv)), dim3(kB), 0, stream, m, dKind, dFirst, dLast, dLeafOrd, dTreeOrd,
                       dNodeBox, dTbox, dTsize);
    int nTreelets = 0, errNow = 0;
    GB_CHECK(hipMemcpyAsync(&nTreelets, dTreeOrd + m, sizeof(int), hipMemcpyDeviceToHost, stream), "read treelet count");
    GB_CHECK(hipMemcpyAsync(&errNow, dErr, sizeof(int), hipMemcpyDeviceToHost, stream), "read error flag");
    GB_CHECK(hipStreamSynchronize(stream), "sync after classify");
    if (errNow == kErrLeafSize) {
        *error = "nnbvh_build_create: more than 65535 primitives share one Morton code (leaf too large)";
        return false;
    }
    if (nTreelets < 1 || nTreelets > 4096) {
        *error = "gpu build: internal error (treelet count)";
        return false;
    }
    std::vector<float> tbox(6 * (size_t)nTreelets);
    std::vector<int> tsize(nTreelets);
    GB_CHECK(hipMemcpy(tbox.data(), dTbox, tbox.size() * sizeof(float), hipMemcpyDeviceToHost), "read treelet bounds");
    GB_CHECK(hipMemcpy(tsize.data(), dTsize, tsize.size() * sizeof(int), hipMemcpyDeviceToHost), "read treelet sizes");
    out->n_treelets = nTreelets;
    out->ms[1] = ms_since(t0);
    if (std::getenv("NNBVH_BUILD_DEBUG")) {
        int nLeaves = 0;
        (void)hipMemcpy(&nLeaves, dLeafOrd + m, sizeof(int), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "gpu build: n %d, distinct codes %d, leaves %d, treelets %d\n", n, m, nLeaves, nTreelets);
        for (int t = 0; t < std::min(nTreelets, 4); ++t)
            std::fprintf(stderr, "  treelet %d: size %d box %g %g %g  %g %g %g\n", t, tsize[t], tbox[6 * t], tbox[6 * t + 1],
                         tbox[6 * t + 2], tbox[6 * t + 3], tbox[6 * t + 4], tbox[6 * t + 5]);
    }
    for (int t = 0; t < nTreelets; ++t) {
        bool fine = tsize[t] >= 1 && (tsize[t] & 1);
        for (int k = 0; k < 6; ++k) fine = fine && std::isfinite(tbox[6 * (size_t)t + k]);
        if (!fine) {
            *error = "nnbvh_build_create: non-finite primitive bounds (or internal error) in treelet " + std::to_string(t);
            return false;
        }
    }

    t0 = std::chrono::steady_clock::now();
    UpperLayout up;
    if (!hlbvh_upper_layout(tbox.data(), tsize.data(), nTreelets, &up, error)) return false;
    out->ms[2] = ms_since(t0);

    t0 = std::chrono::steady_clock::now();
    nnbvh_linear_node *dNodes = mem.get<nnbvh_linear_node>(up.total_nodes);
    nnbvh_prim *dOrdered = mem.get<nnbvh_prim>(n);
    if (!mem.ok) return false;
    GB_CHECK(hipMemcpyAsync(dTbase, up.base.data(), nTreelets * sizeof(int), hipMemcpyHostToDevice, stream), "copy bases");
    GB_CHECK(hipMemcpyAsync(dTdepth, up.depth.data(), nTreelets * sizeof(int), hipMemcpyHostToDevice, stream), "copy depths");
    hipLaunchKernelGGL(k_emit, dim3(grid_all(nv)), dim3(kB), 0, stream, m, dKind, dBit, dPstart, dFirst, dLast, dLeft,
                       dParent, dLeafOrd, dTreeOrd, dTbase, dTdepth, dNodeBox, dNodes, dMaxDepth);
    hipLaunchKernelGGL(k_gather_prims, dim3(grid_for(n)), dim3(kB), 0, stream, dPrims, dIdxS, n, dOrdered);
    GB_CHECK(hipGetLastError(), "kernel launch");
    GB_CHECK(hipStreamSynchronize(stream), "sync after emit");
    out->ms[3] = ms_since(t0);

    t0 = std::chrono::steady_clock::now();
    out->depth = 0;
    GB_CHECK(hipMemcpy(&out->depth, dMaxDepth, sizeof(int), hipMemcpyDeviceToHost), "read depth");
    out->total_nodes = up.total_nodes;
    if (out->keep_on_device) {
        // upper nodes join the device array; nothing else leaves the device
        if (!up.upper_index.empty()) {
            int *dUi = mem.get<int>(up.upper_index.size());
            nnbvh_linear_node *dUn = mem.get<nnbvh_linear_node>(up.upper_nodes.size());
            if (!mem.ok) return false;
            GB_CHECK(hipMemcpy(dUi, up.upper_index.data(), up.upper_index.size() * sizeof(int), hipMemcpyHostToDevice), "copy upper");
            GB_CHECK(hipMemcpy(dUn, up.upper_nodes.data(), up.upper_nodes.size() * sizeof(nnbvh_linear_node), hipMemcpyHostToDevice), "copy upper");
            hipLaunchKernelGGL(k_scatter_nodes, dim3(grid_all((long)up.upper_index.size())), dim3(kB), 0, stream, dUi, dUn,
                               (int)up.upper_index.size(), dNodes);
            GB_CHECK(hipStreamSynchronize(stream), "sync (upper nodes)");
        }
        out->d_nodes = dNodes;
        out->d_ordered = dOrdered;
        out->d_verts = dVerts;
        mem.release(dNodes);
        mem.release(dOrdered);
        mem.release(dVerts);
        out->ms[4] = ms_since(t0);
        return true;
    }
    out->nodes.resize((size_t)up.total_nodes);
    out->ordered.resize((size_t)n);
    GB_CHECK(hipMemcpy(out->nodes.data(), dNodes, out->nodes.size() * sizeof(nnbvh_linear_node), hipMemcpyDeviceToHost),
             "read nodes");
    GB_CHECK(hipMemcpy(out->ordered.data(), dOrdered, out->ordered.size() * sizeof(nnbvh_prim), hipMemcpyDeviceToHost),
             "read ordered prims");
    for (size_t k = 0; k < up.upper_index.size(); ++k) out->nodes[(size_t)up.upper_index[k]] = up.upper_nodes[k];
    out->ms[4] = ms_since(t0);
    return true;
}

// =================================================================================================
// SAH build on the device: buildRecursive's SAH branch (aggregates.cpp:192-387) with the SAME tree
// and the SAME leaf-ordered primitive table as the host builder (bvh_build.cpp), hence as the
// reference.  What makes that possible:
//   * every split decision depends only on the SET of primitives of a node (bounds, centroid bounds,
//     12 bucket counts / bounds: exact min / max / integer sums, order-free) and on float
//     expressions evaluated here operation for operation (sah_choose below is compiled for host
//     and device from one source, -ffp-contract=off);
//   * the ORDER of primitives — which decides the leaf table and every later std::nth_element tie —
//     is what std::partition leaves behind, and libstdc++'s partition has a closed form: elements
//     of the left part that satisfy the predicate and elements of the right part that do not stay
//     where they are; the k-th offender of the left part (from the left) is swapped with the k-th
//     offender of the right part counted from the RIGHT end.  Ranks come from ballots / prefix sums,
//     so a node is partitioned in parallel with the sequential algorithm's exact result;
//   * stored bounds: a leaf's are the in-order fold over its primitives, an interior node's the
//     fold child 0 then child 1 (:371-373) — done after the layout, level by level from the leaves.
// Phase A: nodes with more than kSmallSegment (256) primitives, breadth-first, whole-grid kernels per
//          level (tile = 2048 primitives of one node), the 12-bucket decision on the host (a few
//          thousand nodes in total).
// Phase B: every remaining subtree is built depth-first by ONE wavefront (explicit stack in LDS,
//          nodes written in DFS order into the subtree's pool slice).
// Phase C: DFS layout of the phase-A nodes + subtree slices, copy, bounds, download.
namespace {

constexpr int kSahBuckets = 12;
constexpr int kSmallSegmentDefault = 256;
constexpr int kTile = 2048;

struct HBox {
    float mn[3], mx[3];
};
__host__ __device__ inline void hb_init(HBox &b) {
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = 3.402823466e+38f;
        b.mx[k] = -3.402823466e+38f;
    }
}
__host__ __device__ inline void hb_add(HBox &b, const HBox &o) {  // Union, first of equals kept
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = o.mn[k] < b.mn[k] ? o.mn[k] : b.mn[k];
        b.mx[k] = b.mx[k] < o.mx[k] ? o.mx[k] : b.mx[k];
    }
}
__host__ __device__ inline float hb_area(const HBox &b) {  // util/vecmath.h:1293-1296
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
__host__ __device__ inline int hb_maxdim(const HBox &b) {  // util/vecmath.h:1305-1313
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (dx > dy && dx > dz) return 0;
    else if (dy > dz) return 1;
    else return 2;
}
// which of the 12 buckets a primitive's centroid falls in (:312-317, Bounds3::Offset vecmath.h:1322-1331)
__host__ __device__ inline int sah_bucket(float centroid, float cmn, float cmx) {
    float o = centroid - cmn;
    if (cmx > cmn) o /= cmx - cmn;
    int b = kSahBuckets * o;
    if (b >= kSahBuckets) b = kSahBuckets - 1;  // == 12 for the largest centroid (:316); beyond only for
    if (b < 0) b = 0;                           // non-finite input, which must not index out of range
    return b;
}
// the decision of :319-371 from the node's buckets: best split, how many primitives go left,
// and whether the node is split at all
struct SahChoice {
    int best, mid, split;
};
// bucket bounds arrive as 6 order-preserving keys per bucket (min xyz, max xyz), exactly as the
// atomics left them, in LDS (wavefront subtrees) or host memory (breadth-first phase)
__host__ __device__ inline float sah_key_to_float(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    std::memcpy(&f, &u, 4);
    return f;
#endif
}
__host__ __device__ inline HBox sah_bucket_box(const unsigned *keys, int b) {
    HBox r;
    const unsigned *k6 = keys + 6 * b;
    r.mn[0] = sah_key_to_float(k6[0]);
    r.mn[1] = sah_key_to_float(k6[1]);
    r.mn[2] = sah_key_to_float(k6[2]);
    r.mx[0] = sah_key_to_float(k6[3]);
    r.mx[1] = sah_key_to_float(k6[4]);
    r.mx[2] = sah_key_to_float(k6[5]);
    return r;
}
__host__ __device__ inline SahChoice sah_choose(const int *count, const unsigned *keys, const HBox &bounds, int n,
                                                int maxPrims) {
    constexpr int nSplits = kSahBuckets - 1;
    float costs[nSplits];
    for (int i = 0; i < nSplits; ++i) costs[i] = 0;
    int below = 0;
    HBox bbelow;
    hb_init(bbelow);
#pragma unroll 1
    for (int i = 0; i < nSplits; ++i) {
        hb_add(bbelow, sah_bucket_box(keys, i));
        below += count[i];
        costs[i] += below * hb_area(bbelow);
    }
    int above = 0;
    HBox babove;
    hb_init(babove);
#pragma unroll 1
    for (int i = nSplits; i >= 1; --i) {
        hb_add(babove, sah_bucket_box(keys, i));
        above += count[i];
        costs[i - 1] += above * hb_area(babove);
    }
    int best = -1;
    float minCost = __builtin_inff();
#pragma unroll 1
    for (int i = 0; i < nSplits; ++i)
        if (costs[i] < minCost) {
            minCost = costs[i];
            best = i;
        }
    const float leafCost = (float)n;
    minCost = 1.f / 2.f + minCost / hb_area(bounds);
    SahChoice c;
    c.best = best;
    c.split = (n > maxPrims || minCost < leafCost) ? 1 : 0;
    c.mid = 0;
    for (int i = 0; i <= best; ++i) c.mid += count[i];
    return c;
}

__device__ __forceinline__ float centroid_of(const Box6 &b, int dim) { return .5f * b.mn[dim] + .5f * b.mx[dim]; }

// ---- phase A ---------------------------------------------------------------------------------------
struct Tile {
    int seg, start, n;  // a run of one big node's primitives
};
struct SegInfo {  // per big node of the current level
    int start, n, mid, dim, best;
    float cmn, cmx;  // centroid bounds in `dim`
};

// bounds and centroid bounds of every big node of the level (values; the sign of zeros is not needed)
__global__ __launch_bounds__(kB) void k_seg_reduce(const Tile *__restrict__ tiles, const Box6 *__restrict__ pb,
                                                   const int *__restrict__ perm, unsigned *segAcc) {
    __shared__ float red[12][kB / 64];
    const Tile t = tiles[blockIdx.x];
    float v[12];
    for (int k = 0; k < 3; ++k) {
        v[k] = v[6 + k] = 3.402823466e+38f;
        v[3 + k] = v[9 + k] = -3.402823466e+38f;
    }
    for (int j = threadIdx.x; j < t.n; j += kB) {
        const Box6 b = pb[perm[t.start + j]];
        for (int k = 0; k < 3; ++k) {
            v[k] = fminf(v[k], b.mn[k]);
            v[3 + k] = fmaxf(v[3 + k], b.mx[k]);
            const float c = .5f * b.mn[k] + .5f * b.mx[k];
            v[6 + k] = fminf(v[6 + k], c);
            v[9 + k] = fmaxf(v[9 + k], c);
        }
    }
    for (int q = 0; q < 12; ++q) {
        const bool isMin = (q % 6) < 3;
        for (int off = 32; off >= 1; off >>= 1) {
            const float o = __shfl_xor(v[q], off);
            v[q] = isMin ? fminf(v[q], o) : fmaxf(v[q], o);
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 12; ++q) red[q][threadIdx.x >> 6] = v[q];
    __syncthreads();
    if (threadIdx.x < 12) {
        const int q = threadIdx.x;
        const bool isMin = (q % 6) < 3;
        float r = red[q][0];
        for (int w = 1; w < kB / 64; ++w) r = isMin ? fminf(r, red[q][w]) : fmaxf(r, red[q][w]);
        if (isMin) atomicMin(&segAcc[12 * t.seg + q], f2key(r));
        else atomicMax(&segAcc[12 * t.seg + q], f2key(r));
    }
}

// the 12 buckets of every big node: counts and bounds (:305-318)
__global__ __launch_bounds__(kB) void k_seg_buckets(const Tile *__restrict__ tiles, const SegInfo *__restrict__ info,
                                                    const Box6 *__restrict__ pb, const int *__restrict__ perm,
                                                    unsigned *segKeys, int *segCounts) {
    __shared__ unsigned keys[kSahBuckets * 6];
    __shared__ int counts[kSahBuckets];
    const Tile t = tiles[blockIdx.x];
    const SegInfo si = info[t.seg];
    if (threadIdx.x < kSahBuckets) counts[threadIdx.x] = 0;
    if (threadIdx.x < kSahBuckets * 6) keys[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    for (int j = threadIdx.x; j < t.n; j += kB) {
        const Box6 b = pb[perm[t.start + j]];
        const int bk = sah_bucket(centroid_of(b, si.dim), si.cmn, si.cmx);
        atomicAdd(&counts[bk], 1);
        for (int k = 0; k < 3; ++k) {
            atomicMin(&keys[bk * 6 + k], f2key(b.mn[k]));
            atomicMax(&keys[bk * 6 + 3 + k], f2key(b.mx[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < kSahBuckets && counts[threadIdx.x]) atomicAdd(&segCounts[kSahBuckets * t.seg + threadIdx.x], counts[threadIdx.x]);
    if (threadIdx.x < kSahBuckets * 6 && counts[threadIdx.x / 6]) {
        if ((threadIdx.x % 6) < 3) atomicMin(&segKeys[kSahBuckets * 6 * t.seg + threadIdx.x], keys[threadIdx.x]);
        else atomicMax(&segKeys[kSahBuckets * 6 * t.seg + threadIdx.x], keys[threadIdx.x]);
    }
}

// std::partition, step 1: who is on the wrong side of `mid`
__global__ __launch_bounds__(kB) void k_seg_flags(const Tile *__restrict__ tiles, const SegInfo *__restrict__ info,
                                                  const Box6 *__restrict__ pb, const int *__restrict__ perm,
                                                  int *__restrict__ lfFlag, int *__restrict__ rtFlag) {
    const Tile t = tiles[blockIdx.x];
    const SegInfo si = info[t.seg];
    for (int j = threadIdx.x; j < t.n; j += kB) {
        const int i = t.start + j;
        const bool pred = sah_bucket(centroid_of(pb[perm[i]], si.dim), si.cmn, si.cmx) <= si.best;
        const bool left = i < si.start + si.mid;
        lfFlag[i] = (left && !pred) ? 1 : 0;
        rtFlag[i] = (!left && pred) ? 1 : 0;
    }
}
// step 2: the k-th offender of the left part (from the left) and of the right part (from the right)
__global__ __launch_bounds__(kB) void k_seg_positions(const Tile *__restrict__ tiles, const SegInfo *__restrict__ info,
                                                      const int *__restrict__ lfFlag, const int *__restrict__ lfScan,
                                                      const int *__restrict__ rtFlag, const int *__restrict__ rtScan,
                                                      int *__restrict__ lfPos, int *__restrict__ rtPos) {
    const Tile t = tiles[blockIdx.x];
    const SegInfo si = info[t.seg];
    const int end = si.start + si.n;
    for (int j = threadIdx.x; j < t.n; j += kB) {
        const int i = t.start + j;
        if (lfFlag[i]) lfPos[si.start + (lfScan[i] - lfScan[si.start])] = i;
        if (rtFlag[i]) rtPos[si.start + (rtScan[end] - rtScan[i + 1])] = i;
    }
}
// step 3: swap them pairwise
__global__ __launch_bounds__(kB) void k_seg_swap(const Tile *__restrict__ tiles, const SegInfo *__restrict__ info,
                                                 const int *__restrict__ lfScan, const int *__restrict__ lfPos,
                                                 const int *__restrict__ rtPos, int *perm) {
    const Tile t = tiles[blockIdx.x];
    const SegInfo si = info[t.seg];
    const int nSwaps = lfScan[si.start + si.mid] - lfScan[si.start];
    for (int j = threadIdx.x; j < t.n; j += kB) {
        const int k = t.start + j - si.start;
        if (k < nSwaps) {
            const int a = lfPos[si.start + k], b = rtPos[si.start + k];
            const int va = perm[a], vb = perm[b];
            perm[a] = vb;
            perm[b] = va;
        }
    }
}

// per-node decisions on the device (one thread per big node of the level): nothing but the final
// {dim, mid} verdicts crosses PCIe
struct SegStart {
    int start, n;
};
__global__ __launch_bounds__(kB) void k_seg_init(int nSeg, unsigned *__restrict__ segAcc, unsigned *__restrict__ segKeys,
                                                 int *__restrict__ segCounts) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < 12 * nSeg) segAcc[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    if (i < kSahBuckets * 6 * nSeg) segKeys[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    if (i < kSahBuckets * nSeg) segCounts[i] = 0;
}
// :221, :241-253: flat bounds or coincident centroids -> no split (the node is handed to a wavefront,
// which makes the leaf); otherwise the split axis and the centroid range along it
__global__ __launch_bounds__(kB) void k_seg_prepare(const SegStart *__restrict__ segs, int nSeg,
                                                    const unsigned *__restrict__ segAcc, SegInfo *__restrict__ info,
                                                    HBox *__restrict__ segBounds) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s >= nSeg) return;
    HBox b, cb;
    for (int k = 0; k < 3; ++k) {
        b.mn[k] = sah_key_to_float(segAcc[12 * s + k]);
        b.mx[k] = sah_key_to_float(segAcc[12 * s + 3 + k]);
        cb.mn[k] = sah_key_to_float(segAcc[12 * s + 6 + k]);
        cb.mx[k] = sah_key_to_float(segAcc[12 * s + 9 + k]);
    }
    const int dim = hb_maxdim(cb);
    const bool noSplit = hb_area(b) == 0 || cb.mx[dim] == cb.mn[dim];
    SegInfo si;
    si.start = segs[s].start;
    si.n = segs[s].n;
    si.mid = 0;
    si.dim = dim;
    si.best = noSplit ? -2 : -1;  // -2: decided, no split; -1: buckets wanted
    si.cmn = cb.mn[dim];
    si.cmx = cb.mx[dim];
    info[s] = si;
    segBounds[s] = b;
}
__global__ __launch_bounds__(kB) void k_seg_choose(int nSeg, int maxPrims, const int *__restrict__ segCounts,
                                                   const unsigned *__restrict__ segKeys,
                                                   const HBox *__restrict__ segBounds, SegInfo *info,
                                                   int2 *__restrict__ verdict) {
    const int s = blockIdx.x * kB + threadIdx.x;
    if (s >= nSeg) return;
    SegInfo si = info[s];
    int mid = 0;
    if (si.best != -2) {
        const SahChoice ch = sah_choose(segCounts + kSahBuckets * s, segKeys + kSahBuckets * 6 * s, segBounds[s], si.n,
                                        maxPrims);
        if (ch.split && ch.mid > 0 && ch.mid < si.n) {
            si.best = ch.best;
            mid = ch.mid;
        } else {
            si.best = -1;
        }
    } else {
        si.best = -1;
    }
    si.mid = mid;  // mid == 0: no split; the flags kernel then leaves the node's primitives in place
    info[s] = si;
    verdict[s] = make_int2(si.dim, mid);
}

// ---- phase B: one wavefront builds one subtree ----------------------------------------------------------
struct SmallSeg {
    int start, n, pool;
};
constexpr int kSubtreeStack = 64;

// sah_choose for a wavefront: lane i < 11 prices split i (the same additions in the same order as
// the sequential loops: buckets 0..i ascending for the part below, 11..i+1 descending for the part
// above, cost = (0 + below) + above), then every lane scans the 11 costs in order, first minimum wins.
__device__ __forceinline__ SahChoice sah_choose_wave(int lane, const int *count, const unsigned *keys,
                                                     const HBox &bounds, int n, int maxPrims) {
    float cost = __builtin_inff();
    if (lane < kSahBuckets - 1) {
        HBox acc;
        hb_init(acc);
        int cnt = 0;
#pragma unroll 1
        for (int q = 0; q <= lane; ++q) {
            hb_add(acc, sah_bucket_box(keys, q));
            cnt += count[q];
        }
        float c = 0;
        c += cnt * hb_area(acc);
        hb_init(acc);
        cnt = 0;
#pragma unroll 1
        for (int q = kSahBuckets - 1; q > lane; --q) {
            hb_add(acc, sah_bucket_box(keys, q));
            cnt += count[q];
        }
        c += cnt * hb_area(acc);
        cost = c;
    }
    int best = -1;
    float minCost = __builtin_inff();
#pragma unroll 1
    for (int i = 0; i < kSahBuckets - 1; ++i) {
        const float c = __shfl(cost, i);
        if (c < minCost) {
            minCost = c;
            best = i;
        }
    }
    const float leafCost = (float)n;
    minCost = 1.f / 2.f + minCost / hb_area(bounds);
    SahChoice ch;
    ch.best = best;
    ch.split = (n > maxPrims || minCost < leafCost) ? 1 : 0;
    ch.mid = 0;
#pragma unroll 1
    for (int i = 0; i <= best; ++i) ch.mid += count[i];
    return ch;
}

__global__ __launch_bounds__(64) void k_sah_subtrees(const SmallSeg *__restrict__ segs, int nSegs, int maxPrims,
                                                     const Box6 *__restrict__ pb, int *perm, int *lfPos, int *rtPos,
                                                     nnbvh_linear_node *pool, int *segCount, int *segDepth,
                                                     int *err) {
    __shared__ int stStart[kSubtreeStack], stN[kSubtreeStack], stParent[kSubtreeStack], stDepth[kSubtreeStack];
    __shared__ unsigned bkeys[kSahBuckets * 6];
    __shared__ int bcount[kSahBuckets];
    __shared__ int lpos[64], rpos[64];
    const int s = blockIdx.x;
    if (s >= nSegs) return;
    const int lane = threadIdx.x;
    const SmallSeg sg = segs[s];
    nnbvh_linear_node *nodes = pool + sg.pool;
    int sp = 0, idx = 0, maxDepth = 0;
    int start = sg.start, n = sg.n, depth = 0;
    for (;;) {
        const int me = idx++;
        // A node of at most 64 primitives lives in registers (one primitive per lane) for all of its
        // passes; larger ones loop over their range.
        const bool small = n <= 64;
        const bool mine = lane < n;
        int myIdx = -1;
        Box6 myBox;
        box_init(myBox);
        if (mine) {
            myIdx = perm[start + lane];
            myBox = pb[myIdx];
        }
        // bounds and centroid bounds of the node (values)
        float v[12];
        for (int k = 0; k < 3; ++k) {
            v[k] = v[6 + k] = 3.402823466e+38f;
            v[3 + k] = v[9 + k] = -3.402823466e+38f;
        }
        if (mine)
            for (int k = 0; k < 3; ++k) {
                v[k] = myBox.mn[k];
                v[3 + k] = myBox.mx[k];
                v[6 + k] = v[9 + k] = .5f * myBox.mn[k] + .5f * myBox.mx[k];
            }
        for (int j = lane + 64; j < n; j += 64) {
            const Box6 b = pb[perm[start + j]];
            for (int k = 0; k < 3; ++k) {
                v[k] = fminf(v[k], b.mn[k]);
                v[3 + k] = fmaxf(v[3 + k], b.mx[k]);
                const float c = .5f * b.mn[k] + .5f * b.mx[k];
                v[6 + k] = fminf(v[6 + k], c);
                v[9 + k] = fmaxf(v[9 + k], c);
            }
        }
        // butterfly over the lanes that can hold data (lanes >= n hold the neutral element); most
        // nodes are tiny, so most of the 6 steps are skipped
        int span = 64;
        if (n < 64) {
            span = 1;
            while (span < n) span <<= 1;
        }
        for (int q = 0; q < 12; ++q) {
            const bool isMin = (q % 6) < 3;
            for (int off = 32; off >= 1; off >>= 1) {
                if (off >= span) continue;
                const float o = __shfl_xor(v[q], off);
                v[q] = isMin ? fminf(v[q], o) : fmaxf(v[q], o);
            }
            v[q] = __shfl(v[q], 0);  // one value for the whole wavefront (a zero's sign may differ per lane)
        }
        HBox bounds, cb;
        for (int k = 0; k < 3; ++k) {
            bounds.mn[k] = v[k], bounds.mx[k] = v[3 + k];
            cb.mn[k] = v[6 + k], cb.mx[k] = v[9 + k];
        }
        bool leaf = hb_area(bounds) == 0 || n == 1;  // :221
        int dim = 0, mid = 0, best = 0;
        if (!leaf) {
            dim = hb_maxdim(cb);
            if (cb.mx[dim] == cb.mn[dim]) leaf = true;  // :243
        }
        if (!leaf) {
            const float cmn = cb.mn[dim], cmx = cb.mx[dim];
            if (n <= 2) {  // :289-296: nth_element of two = put the smaller centroid first
                const float c0 = __shfl(centroid_of(myBox, dim), 0), c1 = __shfl(centroid_of(myBox, dim), 1);
                if (c1 < c0 && lane < 2) perm[start + (1 - lane)] = myIdx;
                mid = n / 2;
                __threadfence_block();
            } else {
                if (lane < kSahBuckets) bcount[lane] = 0;
                for (int q = lane; q < kSahBuckets * 6; q += 64) bkeys[q] = (q % 6) < 3 ? 0xffffffffu : 0u;
                __syncthreads();
                int myBucket = 0;
                if (mine) {
                    myBucket = sah_bucket(centroid_of(myBox, dim), cmn, cmx);
                    atomicAdd(&bcount[myBucket], 1);
                    for (int k = 0; k < 3; ++k) {
                        atomicMin(&bkeys[myBucket * 6 + k], f2key(myBox.mn[k]));
                        atomicMax(&bkeys[myBucket * 6 + 3 + k], f2key(myBox.mx[k]));
                    }
                }
                for (int j = lane + 64; j < n; j += 64) {
                    const Box6 b = pb[perm[start + j]];
                    const int bk = sah_bucket(centroid_of(b, dim), cmn, cmx);
                    atomicAdd(&bcount[bk], 1);
                    for (int k = 0; k < 3; ++k) {
                        atomicMin(&bkeys[bk * 6 + k], f2key(b.mn[k]));
                        atomicMax(&bkeys[bk * 6 + 3 + k], f2key(b.mx[k]));
                    }
                }
                __syncthreads();
                const SahChoice ch = sah_choose_wave(lane, bcount, bkeys, bounds, n, maxPrims);
                __syncthreads();  // everyone has read the buckets before the next node clears them
                if (!ch.split) {
                    leaf = true;
                } else if (small) {
                    best = ch.best;
                    mid = ch.mid;
                    // std::partition(bucket <= best) by ranks, in registers: the k-th offender of the
                    // left part (from the left) trades places with the k-th of the right part from the right
                    const bool pred = myBucket <= best;
                    const bool leftOff = mine && lane < mid && !pred, rightOff = mine && lane >= mid && pred;
                    const unsigned long long mL = __ballot(leftOff), mR = __ballot(rightOff);
                    const int kL = __popcll(mL & ((1ull << lane) - 1ull));
                    const int kR = lane == 63 ? 0 : __popcll(mR >> (lane + 1));
                    if (leftOff) lpos[kL] = lane;
                    if (rightOff) rpos[kR] = lane;
                    __syncthreads();
                    if (leftOff) perm[start + rpos[kL]] = myIdx;
                    if (rightOff) perm[start + lpos[kR]] = myIdx;
                    if (__popcll(mL) != __popcll(mR) && lane == 0) *err = 100;  // cannot happen
                    __syncthreads();
                    __threadfence_block();
                } else {
                    best = ch.best;
                    mid = ch.mid;
                    // the same by ranks through scratch: offenders of the left part from the left ...
                    int nl = 0;
                    for (int base = 0; base < mid; base += 64) {
                        const int j = base + lane;
                        bool f = false;
                        if (j < mid) f = sah_bucket(centroid_of(pb[perm[start + j]], dim), cmn, cmx) > best;
                        const unsigned long long mask = __ballot(f);
                        if (f) lfPos[start + nl + __popcll(mask & ((1ull << lane) - 1ull))] = start + j;
                        nl += __popcll(mask);
                    }
                    // ... offenders of the right part from the right
                    int nr = 0;
                    for (int top = n; top > mid; top -= 64) {
                        const int j = top - 1 - lane;
                        bool f = false;
                        if (j >= mid) f = sah_bucket(centroid_of(pb[perm[start + j]], dim), cmn, cmx) <= best;
                        const unsigned long long mask = __ballot(f);
                        if (f) rtPos[start + nr + __popcll(mask & ((1ull << lane) - 1ull))] = start + j;
                        nr += __popcll(mask);
                    }
                    if (nl != nr && lane == 0) *err = 100;  // cannot happen: both equal the number of swaps
                    __threadfence_block();
                    for (int k = lane; k < nl; k += 64) {
                        const int a = lfPos[start + k], b = rtPos[start + k];
                        const int va = perm[a], vb = perm[b];
                        perm[a] = vb;
                        perm[b] = va;
                    }
                    __threadfence_block();
                }
            }
        }
        if (!leaf) {
            if (mid <= 0 || mid >= n) {  // cannot happen (buckets 0 and 11 are never empty): no endless loop
                if (lane == 0) *err = 101;
                leaf = true;
            }
        }
        if (!leaf) {
            if (lane == 0) {
                nnbvh_linear_node nd;
                for (int k = 0; k < 3; ++k) nd.pmin[k] = nd.pmax[k] = 0;
                nd.offset = 0;  // second child: set when it is created
                nd.nprims = 0;
                nd.axis = (uint8_t)dim;
                nd.pad = (uint8_t)depth;  // carried to phase C, cleared there
                nodes[me] = nd;
            }
            if (sp >= kSubtreeStack) {
                if (lane == 0) *err = 102;
                break;
            }
            if (lane == 0) {
                stStart[sp] = start + mid;
                stN[sp] = n - mid;
                stParent[sp] = me;
                stDepth[sp] = depth + 1;
            }
            ++sp;
            n = mid;
            ++depth;
            __syncthreads();
            continue;
        }
        // leaf (:222-236 / :243-253 / :365-369): bounds = in-order fold over its primitives
        {
            Box6 b;
            if (small) {
                b = myBox;  // one primitive per lane, already in order (empty lanes hold the empty box)
            } else {
                const int chunk = (n + 63) / 64;
                const int lo = lane * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
                box_init(b);
                for (int j = lo; j < hi; ++j) box_add(b, pb[perm[start + j]]);
            }
            for (int off = 1; off < (small ? span : 64); off <<= 1) {
                Box6 o;
                for (int k = 0; k < 3; ++k) {
                    o.mn[k] = __shfl_down(b.mn[k], off);
                    o.mx[k] = __shfl_down(b.mx[k], off);
                }
                if (lane + off < 64) box_add(b, o);
            }
            if (lane == 0) {
                nnbvh_linear_node nd;
                for (int k = 0; k < 3; ++k) {
                    nd.pmin[k] = b.mn[k];
                    nd.pmax[k] = b.mx[k];
                }
                nd.offset = start;
                nd.nprims = (uint16_t)n;
                nd.axis = 0;
                nd.pad = (uint8_t)depth;
                nodes[me] = nd;
                if (n > 65535) *err = kErrLeafSize;
            }
            if (depth > maxDepth) maxDepth = depth;
        }
        if (sp == 0) break;
        --sp;
        __syncthreads();
        start = stStart[sp];
        n = stN[sp];
        depth = stDepth[sp];
        if (lane == 0) nodes[stParent[sp]].offset = idx;  // the node about to be created
        __syncthreads();
    }
    if (lane == 0) {
        segCount[s] = idx;
        segDepth[s] = maxDepth;
    }
}

// ---- phase C ----------------------------------------------------------------------------------------------
// subtree slices -> final DFS positions
__global__ __launch_bounds__(kB) void k_sah_emit(const SmallSeg *__restrict__ segs, const int *__restrict__ segCount,
                                                 const int *__restrict__ segBase, const int *__restrict__ segBaseDepth,
                                                 const nnbvh_linear_node *__restrict__ pool,
                                                 nnbvh_linear_node *__restrict__ nodes, unsigned char *__restrict__ depthOf) {
    const int s = blockIdx.x;
    const int cnt = segCount[s], base = segBase[s], bd = segBaseDepth[s];
    const nnbvh_linear_node *src = pool + segs[s].pool;
    for (int i = threadIdx.x; i < cnt; i += kB) {
        nnbvh_linear_node nd = src[i];
        depthOf[base + i] = (unsigned char)(bd + nd.pad);
        nd.pad = 0;
        if (nd.nprims == 0) nd.offset += base;
        nodes[base + i] = nd;
    }
}
struct UpperNode {
    int index, second, axis, depth;
};
__global__ __launch_bounds__(kB) void k_sah_upper(const UpperNode *__restrict__ up, int nUp,
                                                  nnbvh_linear_node *__restrict__ nodes, unsigned char *__restrict__ depthOf) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= nUp) return;
    nnbvh_linear_node nd;
    for (int k = 0; k < 3; ++k) nd.pmin[k] = nd.pmax[k] = 0;
    nd.offset = up[i].second;
    nd.nprims = 0;
    nd.axis = (uint8_t)up[i].axis;
    nd.pad = 0;
    nodes[up[i].index] = nd;
    depthOf[up[i].index] = (unsigned char)up[i].depth;
}
// interior bounds = Union(child 0, child 1) (:371-373), one tree level per launch, deepest first
__global__ __launch_bounds__(kB) void k_sah_level_bounds(int nNodes, int level, const unsigned char *__restrict__ depthOf,
                                                         nnbvh_linear_node *nodes) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= nNodes || depthOf[i] != level || nodes[i].nprims != 0) return;
    const nnbvh_linear_node c0 = nodes[i + 1], c1 = nodes[nodes[i].offset];
    Box6 b, o;
    box_init(b);
    for (int k = 0; k < 3; ++k) o.mn[k] = c0.pmin[k], o.mx[k] = c0.pmax[k];
    box_add(b, o);
    for (int k = 0; k < 3; ++k) o.mn[k] = c1.pmin[k], o.mx[k] = c1.pmax[k];
    box_add(b, o);
    for (int k = 0; k < 3; ++k) nodes[i].pmin[k] = b.mn[k], nodes[i].pmax[k] = b.mx[k];
}

struct HostNode {  // a phase-A node
    int child[2];  // >= 0: HostNode index; < 0: ~(small segment index)
    int axis;
};

}  // namespace

bool gpu_sah(const nnbvh_prim *prims, int n, const float *verts, int n_verts, const float *prim_bounds,
             int max_prims_in_node, int device, GpuBuildResult *out, std::string *error) {
    int prev = 0;
    GB_CHECK(hipGetDevice(&prev), "hipGetDevice");
    GB_CHECK(hipSetDevice(device), "hipSetDevice");
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore{prev};
    const int maxPrims = std::min(255, max_prims_in_node);
    int kSmallSegment = kSmallSegmentDefault;  // speed only: where breadth-first hands over to wavefronts
    if (const char *e = std::getenv("NNBVH_SAH_SMALL")) kSmallSegment = std::max(2, std::atoi(e));
    hipStream_t stream = nullptr;
    DevMem mem;
    mem.error = error;

    auto t0 = std::chrono::steady_clock::now();
    nnbvh_prim *dPrims = mem.get<nnbvh_prim>(n);
    float *dVerts = mem.get<float>(3 * (size_t)n_verts);
    float *dCaller = prim_bounds ? mem.get<float>(6 * (size_t)n) : nullptr;
    Box6 *dPb = mem.get<Box6>(n);
    int *dPerm = mem.get<int>(n);
    int *dLfFlag = mem.get<int>((size_t)n + 1), *dRtFlag = mem.get<int>((size_t)n + 1);
    int *dLfScan = mem.get<int>((size_t)n + 1), *dRtScan = mem.get<int>((size_t)n + 1);
    int *dLfPos = mem.get<int>(n), *dRtPos = mem.get<int>(n);
    unsigned *dCodesUnused = mem.get<unsigned>(n);
    int *dScalars = mem.get<int>(16);
    if (!mem.ok) return false;
    GB_CHECK(hipMemcpyAsync(dPrims, prims, (size_t)n * sizeof(nnbvh_prim), hipMemcpyHostToDevice, stream), "copy prims");
    GB_CHECK(hipMemcpyAsync(dVerts, verts, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, stream), "copy verts");
    if (dCaller)
        GB_CHECK(hipMemcpyAsync(dCaller, prim_bounds, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream), "copy bounds");
    const int init[16] = {-1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    GB_CHECK(hipMemcpyAsync(dScalars, init, sizeof init, hipMemcpyHostToDevice, stream), "init scalars");
    GB_CHECK(hipStreamSynchronize(stream), "sync after upload");
    out->ms[0] = ms_since(t0);

    t0 = std::chrono::steady_clock::now();
    int *dErr = dScalars + 6;
    hipLaunchKernelGGL(k_prim_bounds, dim3(grid_for(n, 1024)), dim3(kB), 0, stream, dPrims, dVerts, n_verts, dCaller, n,
                       dPb, (unsigned *)dScalars, dErr);
    hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(kB), 0, stream, dPb, (unsigned *)dScalars, n, dCodesUnused,
                       dPerm);  // only for perm[i] = i
    size_t scanBytes = 0;
    GB_CHECK(rocprim::exclusive_scan(nullptr, scanBytes, dLfFlag, dLfScan, 0, (size_t)n + 1, rocprim::plus<int>(), stream),
             "scan (size query)");
    void *dTmp = mem.get<char>(scanBytes);
    if (!mem.ok) return false;
    int errNow = 0;
    GB_CHECK(hipMemcpyAsync(&errNow, dErr, sizeof(int), hipMemcpyDeviceToHost, stream), "read error flag");
    GB_CHECK(hipStreamSynchronize(stream), "sync after bounds");
    if (errNow != 0) {
        *error = errNow == kErrVertex       ? "nnbvh_build_create: vertex index out of range"
                 : errNow == kErrNeedBounds ? "nnbvh_build_create: instance / host primitives need prim_bounds"
                 : errNow == kErrNonFinite  ? "nnbvh_build_create: non-finite vertex or primitive bounds"
                                            : "nnbvh_build_create: unknown primitive kind";
        return false;
    }

    // ---- phase A: big nodes, breadth-first -------------------------------------------------------------
    struct Big {
        int start, n, host;  // host = HostNode index
    };
    std::vector<HostNode> hostNodes;
    std::vector<SmallSeg> smallSegs;
    std::vector<Big> level;
    int rootRef;  // >= 0 host node, < 0 ~small segment
    auto make_child = [&](int start, int cnt, std::vector<Big> &next) -> int {
        if (cnt > kSmallSegment) {
            hostNodes.push_back(HostNode{{0, 0}, 0});
            next.push_back(Big{start, cnt, (int)hostNodes.size() - 1});
            return (int)hostNodes.size() - 1;
        }
        smallSegs.push_back(SmallSeg{start, cnt, 0});
        return ~((int)smallSegs.size() - 1);
    };
    rootRef = make_child(0, n, level);
    std::vector<Tile> tiles;
    Tile *dTiles = nullptr;
    SegInfo *dInfo = nullptr;
    unsigned *dAcc = nullptr, *dBKeys = nullptr;
    int *dBCounts = nullptr;
    size_t capTiles = 0, capSegs = 0;
    std::vector<SegStart> segStarts;
    std::vector<int2> verdicts;
    SegStart *dSegStart = nullptr;
    HBox *dSegBounds = nullptr;
    int2 *dVerdict = nullptr;
    while (!level.empty()) {
        const int nSeg = (int)level.size();
        tiles.clear();
        segStarts.resize((size_t)nSeg);
        for (int s = 0; s < nSeg; ++s) {
            segStarts[(size_t)s] = SegStart{level[s].start, level[s].n};
            for (int off = 0; off < level[s].n; off += kTile)
                tiles.push_back(Tile{s, level[s].start + off, std::min(kTile, level[s].n - off)});
        }
        if (tiles.size() > capTiles) {
            capTiles = tiles.size() * 2;
            dTiles = mem.get<Tile>(capTiles);
        }
        if ((size_t)nSeg > capSegs) {
            capSegs = (size_t)nSeg * 2;
            dInfo = mem.get<SegInfo>(capSegs);
            dAcc = mem.get<unsigned>(12 * capSegs);
            dBKeys = mem.get<unsigned>(kSahBuckets * 6 * capSegs);
            dBCounts = mem.get<int>(kSahBuckets * capSegs);
            dSegStart = mem.get<SegStart>(capSegs);
            dSegBounds = mem.get<HBox>(capSegs);
            dVerdict = mem.get<int2>(capSegs);
        }
        if (!mem.ok) return false;
        const int nTiles = (int)tiles.size();
        GB_CHECK(hipMemcpyAsync(dTiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice, stream), "copy tiles");
        GB_CHECK(hipMemcpyAsync(dSegStart, segStarts.data(), segStarts.size() * sizeof(SegStart), hipMemcpyHostToDevice, stream),
                 "copy nodes");
        hipLaunchKernelGGL(k_seg_init, dim3(grid_all((long)kSahBuckets * 6 * nSeg)), dim3(kB), 0, stream, nSeg, dAcc, dBKeys,
                           dBCounts);
        hipLaunchKernelGGL(k_seg_reduce, dim3(nTiles), dim3(kB), 0, stream, dTiles, dPb, dPerm, dAcc);
        hipLaunchKernelGGL(k_seg_prepare, dim3(grid_all(nSeg)), dim3(kB), 0, stream, dSegStart, nSeg, dAcc, dInfo, dSegBounds);
        hipLaunchKernelGGL(k_seg_buckets, dim3(nTiles), dim3(kB), 0, stream, dTiles, dInfo, dPb, dPerm, dBKeys, dBCounts);
        hipLaunchKernelGGL(k_seg_choose, dim3(grid_all(nSeg)), dim3(kB), 0, stream, nSeg, maxPrims, dBCounts, dBKeys, dSegBounds,
                           dInfo, dVerdict);
        GB_CHECK(hipMemsetAsync(dLfFlag, 0, ((size_t)n + 1) * sizeof(int), stream), "memset");
        GB_CHECK(hipMemsetAsync(dRtFlag, 0, ((size_t)n + 1) * sizeof(int), stream), "memset");
        hipLaunchKernelGGL(k_seg_flags, dim3(nTiles), dim3(kB), 0, stream, dTiles, dInfo, dPb, dPerm, dLfFlag, dRtFlag);
        GB_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dLfFlag, dLfScan, 0, (size_t)n + 1, rocprim::plus<int>(), stream), "scan");
        GB_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dRtFlag, dRtScan, 0, (size_t)n + 1, rocprim::plus<int>(), stream), "scan");
        hipLaunchKernelGGL(k_seg_positions, dim3(nTiles), dim3(kB), 0, stream, dTiles, dInfo, dLfFlag, dLfScan, dRtFlag,
                           dRtScan, dLfPos, dRtPos);
        hipLaunchKernelGGL(k_seg_swap, dim3(nTiles), dim3(kB), 0, stream, dTiles, dInfo, dLfScan, dLfPos, dRtPos, dPerm);
        verdicts.resize((size_t)nSeg);
        GB_CHECK(hipMemcpyAsync(verdicts.data(), dVerdict, (size_t)nSeg * sizeof(int2), hipMemcpyDeviceToHost, stream), "read verdicts");
        GB_CHECK(hipStreamSynchronize(stream), "sync (level)");
        std::vector<Big> next;
        for (int s = 0; s < nSeg; ++s) {
            const int dim = verdicts[(size_t)s].x, mid = verdicts[(size_t)s].y;
            const int host = level[s].host;
            if (mid <= 0) {
                // a big node that does not split (coincident centroids / flat bounds): it becomes a
                // subtree for the wavefront builder, which reaches the same verdict and makes the leaf
                smallSegs.push_back(SmallSeg{level[s].start, level[s].n, 0});
                hostNodes[(size_t)host].axis = -1;
                hostNodes[(size_t)host].child[0] = ~((int)smallSegs.size() - 1);
                hostNodes[(size_t)host].child[1] = 0;
                continue;
            }
            const int c0 = make_child(level[s].start, mid, next);
            const int c1 = make_child(level[s].start + mid, level[s].n - mid, next);
            hostNodes[(size_t)host].axis = dim;  // (index again: make_child may have grown the vector)
            hostNodes[(size_t)host].child[0] = c0;
            hostNodes[(size_t)host].child[1] = c1;
        }
        level.swap(next);
    }
    // k_seg_flags marks delegated nodes' primitives with pred = (bucket <= -1) = false and mid = 0:
    // every one of them is "right part, predicate false" -> stays in place.
    out->ms[1] = ms_since(t0);

    // ---- phase B ------------------------------------------------------------------------------------------
    t0 = std::chrono::steady_clock::now();
    const int nSmall = (int)smallSegs.size();
    long poolTotal = 0;
    for (SmallSeg &sg : smallSegs) {
        sg.pool = (int)poolTotal;
        poolTotal += 2L * sg.n - 1;
    }
    if (poolTotal >= 0x7fffffffL) {
        *error = "gpu build: too many nodes";
        return false;
    }
    SmallSeg *dSegs = mem.get<SmallSeg>(nSmall);
    nnbvh_linear_node *dPool = mem.get<nnbvh_linear_node>((size_t)poolTotal);
    int *dSegCount = mem.get<int>(nSmall), *dSegDepth = mem.get<int>(nSmall);
    if (!mem.ok) return false;
    GB_CHECK(hipMemcpyAsync(dSegs, smallSegs.data(), (size_t)nSmall * sizeof(SmallSeg), hipMemcpyHostToDevice, stream), "copy subtrees");
    hipLaunchKernelGGL(k_sah_subtrees, dim3(nSmall), dim3(64), 0, stream, dSegs, nSmall, maxPrims, dPb, dPerm, dLfPos,
                       dRtPos, dPool, dSegCount, dSegDepth, dErr);
    std::vector<int> segCount((size_t)nSmall), segDepth((size_t)nSmall);
    GB_CHECK(hipMemcpyAsync(segCount.data(), dSegCount, (size_t)nSmall * sizeof(int), hipMemcpyDeviceToHost, stream), "read counts");
    GB_CHECK(hipMemcpyAsync(segDepth.data(), dSegDepth, (size_t)nSmall * sizeof(int), hipMemcpyDeviceToHost, stream), "read depths");
    GB_CHECK(hipMemcpyAsync(&errNow, dErr, sizeof(int), hipMemcpyDeviceToHost, stream), "read error flag");
    GB_CHECK(hipStreamSynchronize(stream), "sync (subtrees)");
    if (errNow != 0) {
        *error = errNow == kErrLeafSize ? "nnbvh_build_create: a leaf would hold more than 65535 primitives"
                                        : "gpu build: internal error " + std::to_string(errNow);
        return false;
    }
    out->ms[2] = ms_since(t0);

    // ---- phase C: DFS layout (flattenBVH, :505-522) ----------------------------------------------------------
    t0 = std::chrono::steady_clock::now();
    std::vector<int> segBase((size_t)nSmall), segBaseDepth((size_t)nSmall);
    std::vector<UpperNode> upper;
    int offset = 0, maxDepth = 0;
    // iterative DFS over the phase-A tree
    struct Frame {
        int ref, depth, parentUpper, which;
    };
    std::vector<Frame> stack;
    stack.push_back(Frame{rootRef, 0, -1, 0});
    while (!stack.empty()) {
        Frame f = stack.back();
        stack.pop_back();
        int ref = f.ref;
        // a delegated big node stands for its subtree
        if (ref >= 0 && hostNodes[(size_t)ref].axis < 0) ref = hostNodes[(size_t)ref].child[0];
        if (f.which == 1) upper[(size_t)f.parentUpper].second = offset;
        if (ref < 0) {
            const int sidx = ~ref;
            segBase[(size_t)sidx] = offset;
            segBaseDepth[(size_t)sidx] = f.depth;
            offset += segCount[(size_t)sidx];
            maxDepth = std::max(maxDepth, f.depth + segDepth[(size_t)sidx]);
            continue;
        }
        const HostNode &hn = hostNodes[(size_t)ref];
        upper.push_back(UpperNode{offset, 0, hn.axis, f.depth});
        const int me = (int)upper.size() - 1;
        ++offset;
        stack.push_back(Frame{hn.child[1], f.depth + 1, me, 1});  // popped after the whole left subtree
        stack.push_back(Frame{hn.child[0], f.depth + 1, me, 0});
    }
    const int totalNodes = offset;
    if (maxDepth > 255) {
        *error = "gpu build: tree deeper than 255 levels";
        return false;
    }
    nnbvh_linear_node *dNodes = mem.get<nnbvh_linear_node>((size_t)totalNodes);
    unsigned char *dDepthOf = mem.get<unsigned char>((size_t)totalNodes);
    int *dSegBase = mem.get<int>(nSmall), *dSegBaseDepth = mem.get<int>(nSmall);
    UpperNode *dUpper = mem.get<UpperNode>(upper.size());
    nnbvh_prim *dOrdered = mem.get<nnbvh_prim>(n);
    if (!mem.ok) return false;
    GB_CHECK(hipMemcpyAsync(dSegBase, segBase.data(), (size_t)nSmall * sizeof(int), hipMemcpyHostToDevice, stream), "copy bases");
    GB_CHECK(hipMemcpyAsync(dSegBaseDepth, segBaseDepth.data(), (size_t)nSmall * sizeof(int), hipMemcpyHostToDevice, stream), "copy depths");
    if (!upper.empty())
        GB_CHECK(hipMemcpyAsync(dUpper, upper.data(), upper.size() * sizeof(UpperNode), hipMemcpyHostToDevice, stream), "copy upper nodes");
    hipLaunchKernelGGL(k_sah_emit, dim3(nSmall), dim3(kB), 0, stream, dSegs, dSegCount, dSegBase, dSegBaseDepth, dPool, dNodes,
                       dDepthOf);
    if (!upper.empty())
        hipLaunchKernelGGL(k_sah_upper, dim3(grid_all((long)upper.size())), dim3(kB), 0, stream, dUpper, (int)upper.size(),
                           dNodes, dDepthOf);
    for (int lvl = maxDepth - 1; lvl >= 0; --lvl)
        hipLaunchKernelGGL(k_sah_level_bounds, dim3(grid_all(totalNodes)), dim3(kB), 0, stream, totalNodes, lvl, dDepthOf, dNodes);
    hipLaunchKernelGGL(k_gather_prims, dim3(grid_for(n)), dim3(kB), 0, stream, dPrims, dPerm, n, dOrdered);
    GB_CHECK(hipGetLastError(), "kernel launch");
    GB_CHECK(hipStreamSynchronize(stream), "sync after emit");
    out->ms[3] = ms_since(t0);

    t0 = std::chrono::steady_clock::now();
    out->total_nodes = totalNodes;
    if (out->keep_on_device) {
        out->d_nodes = dNodes;
        out->d_ordered = dOrdered;
        out->d_verts = dVerts;
        mem.release(dNodes);
        mem.release(dOrdered);
        mem.release(dVerts);
    } else {
        out->nodes.resize((size_t)totalNodes);
        out->ordered.resize((size_t)n);
        GB_CHECK(hipMemcpy(out->nodes.data(), dNodes, out->nodes.size() * sizeof(nnbvh_linear_node), hipMemcpyDeviceToHost), "read nodes");
        GB_CHECK(hipMemcpy(out->ordered.data(), dOrdered, out->ordered.size() * sizeof(nnbvh_prim), hipMemcpyDeviceToHost), "read ordered prims");
    }
    out->depth = maxDepth;
    out->n_treelets = nSmall;
    out->n_unique_codes = (int)upper.size();
    out->ms[4] = ms_since(t0);
    return true;
}

}  // namespace nnbvh
