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

enum : int { kErrVertex = 1, kErrNeedBounds = 2, kErrKind = 3, kErrLeafSize = 4 };
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
        const int nv = p.kind == NNBVH_PRIM_TRIANGLE ? 3 : (p.kind == NNBVH_PRIM_BILINEAR_PATCH ? 4 : 0);
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
    ~DevMem() {
        for (void *p : ptrs) (void)hipFree(p);
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
    out->nodes.resize((size_t)up.total_nodes);
    out->ordered.resize((size_t)n);
    GB_CHECK(hipMemcpy(out->nodes.data(), dNodes, out->nodes.size() * sizeof(nnbvh_linear_node), hipMemcpyDeviceToHost),
             "read nodes");
    GB_CHECK(hipMemcpy(out->ordered.data(), dOrdered, out->ordered.size() * sizeof(nnbvh_prim), hipMemcpyDeviceToHost),
             "read ordered prims");
    GB_CHECK(hipMemcpy(&out->depth, dMaxDepth, sizeof(int), hipMemcpyDeviceToHost), "read depth");
    for (size_t k = 0; k < up.upper_index.size(); ++k) out->nodes[(size_t)up.upper_index[k]] = up.upper_nodes[k];
    out->ms[4] = ms_since(t0);
    return true;
}

}  // namespace nnbvh
