// kd_trace.hip — KdTreeAggregate::Intersect / IntersectP on gfx950, plus their C ABI.
//
// What is computed (reference: /root/reference/src/pbrt/cpu/aggregates.cpp):
//   closest hit   KdTreeAggregate::Intersect    :973-1067
//   any hit       KdTreeAggregate::IntersectP   :1069-1150
//   root interval Bounds3::IntersectP(o, d, tMax, &t0, &t1)   util/vecmath.h:1547-1571
//   leaves        the same Triangle / BilinearPatch tests as the BVH kernels (trace_math.h)
// Results (hit primitive, t, barycentrics, kdNodesVisited and nTriTests per ray) are bit-identical
// to that code.  HOW it runs is the BVH kernels' scheme: persistent 64-lane
// wavefronts over XCD-aware ray queues, a per-lane state machine whose step kind (node step /
// primitive step / refill) the wave picks from ballots, and the KdNodeToVisit{node, tMin, tMax}
// stack (aggregates.cpp:747-750) as a ring window in LDS that spills its oldest entry to a
// coalesced HBM array.  A node step reads ONE 8-B KdTreeNode; there is no box test below the root
// (a kd-tree clips the ray interval against split planes instead), so node steps are ~4x cheaper
// than a BVH interior step and the kernel is bound by the dependent 8-B loads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nnbvh.h"
#include "nnbvh_internal.h"
#include "trace_math.h"
#include "spawn_math.h"

namespace nnbvh {

constexpr int kKdBlock = 256;
constexpr int kKdW = 8;              // LDS window of the to-visit stack, entries per lane (patch instances)
constexpr int kKdWLean = 4;          // ... of the lean instances: 18 KiB of LDS per block, 8 blocks per CU (measured
                                     // against 8 entries / 5 blocks: crown +8 %, bathroom +11 %)
constexpr int kKdQueues = 8;         // one ray queue per XCD
constexpr int kKdQueueStride = 32;   // words: every queue head on its own 128-B line
constexpr int kKdDone = -1;          // lane carries no ray
constexpr int kKdLeaf = -2;          // lane is walking the primitives of a leaf

struct KdParams {
    const uint2 *nodes;           // KdTreeNode[], {split | index, flags}
    const int32_t *primIndices;   // KdTreeAggregate::primitiveIndices
    const float4 *prims;          // 4 slots of 16 B per primitive, in the caller's primitive order:
                                  // {p0, id} {p1, flags} {p2, 0} {p3 (patch), 0}
    float bmin[3], bmax[3];       // KdTreeAggregate::bounds
    const nnbvh_ray *rays;
    nnbvh_hit *hits;              // MODE 0
    uint8_t *occluded;            // MODE 1
    int32_t *visitedOut, *testsOut;  // MODE 1, nullable
    long n;
    unsigned *queue;
    int nQueues;
    int primWeight, refillWeight, nodeRepeat;
    int hasHostPrims;
    float4 *spill;                // [kMaxStack][grid threads] overflow of the LDS window
    const float4 *extras;         // ATTR instances: 6 slots per primitive {n0} {n1} {n2} {n3} {uv00, uv10} {uv01, uv11}
                                  // (per-vertex normals / uvs of the alpha-tested kinds that read them), else null
};

// util/vecmath.h:1547-1571 with invRayDir = 1 / d[i] taken from the ray's precomputed reciprocals
// (the same division)
DEV bool kd_root_interval(const float bmin[3], const float bmax[3], V3 o, V3 inv, float tMaxRay, float &t0Out,
                          float &t1Out) {
    float t0 = 0.0f, t1 = tMaxRay;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float tNear = (bmin[i] - oo[i]) * ii[i];
        float tFar = (bmax[i] - oo[i]) * ii[i];
        if (tNear > tFar) {
            const float s = tNear;
            tNear = tFar;
            tFar = s;
        }
        tFar *= 1.0f + 2.0f * gamma_f(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    t0Out = t0;
    t1Out = t1;
    return true;
}

// MODE 0: closest hit; MODE 1: any hit (counts written when asked for).
// PATCH = 0: the scene holds no bilinear patches — nothing reads the ray direction after the ray is
// fetched except `ray.d[axis] <= 0` at interior nodes, which rides as three bits beside the shear's
// kz; the patch test, the largest register consumer, is not compiled in.
// W: entries per lane of the LDS window of the to-visit stack.
// O32: nodes, primitive records and primitiveIndices are each below 4 GiB and are fetched through 32-bit
// byte offsets from a scalar base (no 64-bit shift / add per fetch).
// ATTR = 1 (with PATCH = 1): the scene holds alpha-tested triangles of smooth meshes or alpha-tested bilinear patches,
// whose re-trace reads per-vertex attributes from p.extras (bvh_trace.hip's ALPHA = 1 smooth / ALPHA = 2 code)
template <int MODE, int PATCH, int W, int O32, int ATTR = 0>
__global__ __launch_bounds__(kKdBlock, PATCH ? 1 : 2) void kd_trace_kernel(KdParams p) {
    auto at = [](const auto *base, int index) {  // &base[index]
        using T = decltype(base);
        return O32 ? reinterpret_cast<T>(reinterpret_cast<const char *>(base) + (unsigned)index * (unsigned)sizeof(*base))
                   : base + (long)index;
    };
    // KdNodeToVisit {node, tMin, tMax}: the three words of an entry 64 dwords apart
    __shared__ float s_stack[kKdBlock / 64][W][3][64];
    // cold per-ray state ([field][lane]): ray index, best hit (closest), reached-a-host-primitive flag
    constexpr int kColdRi = 0, kColdHit = 1, kColdHost = (MODE == 0) ? 5 : 1, kColdFields = kColdHost + 1;
    __shared__ float s_cold[kKdBlock / 64][kColdFields][64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gtid = blockIdx.x * kKdBlock + threadIdx.x;
    float(*stk)[3][64] = s_stack[wave];
    float(*cold)[64] = s_cold[wave];
    cold[kColdRi][lane] = __int_as_float(-1);
    const long spillStride = (long)gridDim.x * kKdBlock;

    int q = 0;
    if (p.nQueues > 1) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        q = (int)(xcc & 0xf) % p.nQueues;
    }
    int queuesTried = 0;
    const long nRays = p.n;

    RayState r;          // o, 1/d, shear; r.kz also carries bit 4 + k = (d[k] <= 0), see the refill
    V3 d = {0, 0, 0};    // PATCH only: the patch test reads the direction
    float rayTMax = 0.0f, tMin = 0.0f, tMax = 0.0f;
    int visited = 0, tests = 0;
    int cur = kKdDone, sp = 0, base = 0;
    // a lane inside a leaf (cur == kKdLeaf): leafIdx = the primitive to test next, leafPos = where the
    // index after it sits in primitiveIndices, leafLeft = primitives still to test including leafIdx
    int leafIdx = 0, leafPos = 0, leafLeft = 0;
    bool found = false, exhausted = false;

    // aggregates.cpp:1053-1060 / :1098-1105: next entry of the to-visit list, or the ray is finished
    auto pop_or_done = [&]() {
        if (sp > 0) {
            --sp;
            float node = stk[sp & (W - 1)][0][lane];
            float a = stk[sp & (W - 1)][1][lane], b = stk[sp & (W - 1)][2][lane];
            asm volatile("" : "+v"(node), "+v"(a), "+v"(b));
            if (sp < base) {  // rare: the entry lives in the HBM spill array
                const float4 e = p.spill[(long)sp * spillStride + gtid];
                base = sp;
                node = e.x;
                a = e.y;
                b = e.z;
                asm volatile("" : "+v"(node), "+v"(a), "+v"(b));
            }
            cur = __float_as_int(node);
            tMin = a;
            tMax = b;
        } else {
            cur = kKdDone;
        }
    };

    for (;;) {
        const bool isNode = cur >= 0;
        const bool isIdle = cur == kKdDone;
        const int nNode = __popcll(__ballot(isNode));
        const unsigned long long idleMask = __ballot(isIdle);
        const int nIdle = __popcll(idleMask);
        const int nPrim = 64 - nNode - nIdle;
        const int sI = nNode * 16, sP = nPrim * p.primWeight;
        const int sR = exhausted ? 0 : nIdle * p.refillWeight;

        if (nIdle == 64 || (sR > sI && sR > sP)) {
            // ---- retire finished rays, refill idle lanes ------------------------------------
            const int ri = isIdle ? __float_as_int(cold[kColdRi][lane]) : -1;
            if (ri >= 0) {
                const bool needHost = p.hasHostPrims && cold[kColdHost][lane] != 0.0f;
                if (MODE == 0) {
                    float4 h0, h1;
                    h0.x = cold[kColdHit][lane];
                    h0.y = rayTMax;
                    h0.z = cold[kColdHit + 1][lane];
                    h0.w = cold[kColdHit + 2][lane];
                    h1.x = cold[kColdHit + 3][lane];
                    h1.y = __int_as_float(visited);
                    h1.z = __int_as_float(tests);
                    h1.w = needHost ? __int_as_float(-1) : 0.0f;
                    float4 *out = reinterpret_cast<float4 *>(p.hits) + 2 * (long)ri;
                    out[0] = h0;
                    out[1] = h1;
                } else {
                    p.occluded[ri] = found ? 1 : (needHost ? 2 : 0);
                    if (p.visitedOut) p.visitedOut[ri] = visited;
                    if (p.testsOut) p.testsOut[ri] = tests;
                }
            }
            int newRi = -1;
            if (exhausted) break;  // only reached with every lane idle
            for (;;) {
                static_assert(kKdQueues == 8, "queue ranges are computed with a shift by 3");
                const int qShift = p.nQueues > 1 ? 3 : 0;  // nQueues is 1 or 8: a shift, not a 64-bit division
                const long qBegin = (nRays * q) >> qShift, qEnd = (nRays * (q + 1)) >> qShift;
                unsigned got = 0;
                if (lane == 0) got = atomicAdd(&p.queue[q * kKdQueueStride], (unsigned)nIdle);
                got = __builtin_amdgcn_readfirstlane(got);
                const long start = qBegin + (long)got;
                if (start < qEnd) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idleMask >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((unsigned)idleMask, 0u));
                    if (isIdle && start + rank < qEnd) newRi = (int)(start + rank);
                    break;
                }
                if (++queuesTried >= p.nQueues) {
                    exhausted = true;
                    break;
                }
                q = (q + 1 == p.nQueues) ? 0 : q + 1;
            }
            if (isIdle) cold[kColdRi][lane] = __int_as_float(newRi);
            if (newRi >= 0) {
                const float4 *in = reinterpret_cast<const float4 *>(p.rays) + 2 * (long)newRi;
                const float4 r0 = in[0], r1 = in[1];
                r.o = {r0.x, r0.y, r0.z};
                rayTMax = r0.w;
                const V3 dir = {r1.x, r1.y, r1.z};
                if (PATCH) d = dir;
                r.inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};  // aggregates.cpp:980
                ray_shear(r, dir);
                // `ray.d[axis] <= 0` of aggregates.cpp:1002-1003, once per ray
                r.kz |= (dir.x <= 0.0f ? 16 : 0) | (dir.y <= 0.0f ? 32 : 0) | (dir.z <= 0.0f ? 64 : 0);
                if (MODE == 0) {
                    cold[kColdHit][lane] = __int_as_float(-1);
                    cold[kColdHit + 1][lane] = 0.0f;
                    cold[kColdHit + 2][lane] = 0.0f;
                    cold[kColdHit + 3][lane] = 0.0f;
                }
                if (p.hasHostPrims) cold[kColdHost][lane] = 0.0f;
                visited = 0;
                tests = 0;
                found = false;
                sp = base = 0;
                // :975-977: rays that miss the tree's bounds return before anything is counted
                cur = kd_root_interval(p.bmin, p.bmax, r.o, r.inv, rayTMax, tMin, tMax) ? 0 : kKdDone;
            }
            continue;
        }

        if (sP > sI || nNode == 0) {
            // ---- primitive step: lanes inside a leaf test ONE primitive ------------------------
            if (cur == kKdLeaf) {
                // one trip to memory: the primitive's three slots and — if the leaf goes on — the index of
                // the primitive after it, all issued before anything is looked at
                const float4 *rec = at(p.prims, 4 * leafIdx);
                float4 s0 = rec[0], s1 = rec[1], s2 = rec[2];
                int nextIdx = 0;
                if (leafLeft > 1) nextIdx = *at(p.primIndices, leafPos);
                asm volatile("" : "+v"(s0.x), "+v"(s0.y), "+v"(s0.z), "+v"(s0.w), "+v"(s1.x), "+v"(s1.y),
                                  "+v"(s1.z), "+v"(s1.w), "+v"(s2.x), "+v"(s2.y), "+v"(s2.z), "+v"(s2.w), "+v"(nextIdx));
                const unsigned flags = __float_as_uint(s1.w);
                if (flags & kPrimHost) {
                    cold[kColdHost][lane] = 1.0f;
                } else {
                    tests += 1;
                    bool hit;
                    float x0, x1, x2, th;
                    if (!PATCH || !(flags & kPrimPatch)) {
                        hit = triangle_test(r, rayTMax, (flags & kPrimDegenerate) != 0, {s0.x, s0.y, s0.z},
                                            {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z}, x0, x1, x2, th);
                        if (PATCH && hit && (flags & kPrimAlpha)) {
                            // GeometricPrimitive::Intersect / IntersectP with a constant alpha (cpu/primitive.cpp:
                            // 57-70, 79-81), as in the BVH kernels; scenes with such primitives run the PATCH
                            // instances, which keep the ray direction
                            const float a = s2.w;
                            if (a < 1) {
                                const float u = (a <= 0) ? 1.f : hash_float_6f(r.o, d);
                                if (u > a) {
                                    hit = false;
                                    RayState rn = r;
                                    if (ATTR && (flags & kPrimSmooth)) {  // FaceForward(n, ns): shapes.h:939-951
                                        const float4 *ex = p.extras + 6 * (long)leafIdx;
                                        const float4 m0 = ex[0], m1 = ex[1], m2 = ex[2];
                                        rn.o = alpha_retrace_origin({s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z},
                                                                    x0, x1, x2, (flags & kPrimFlipN) != 0, d, true,
                                                                    {m0.x, m0.y, m0.z}, {m1.x, m1.y, m1.z}, {m2.x, m2.y, m2.z});
                                    } else {
                                        rn.o = alpha_retrace_origin({s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z},
                                                                    x0, x1, x2, (flags & kPrimFlipN) != 0, d);
                                    }
                                    tests += 1;  // Triangle::Intersect counts the re-test too
                                    float y0, y1, y2, tn;
                                    if (triangle_test(rn, rayTMax - th, (flags & kPrimDegenerate) != 0, {s0.x, s0.y, s0.z},
                                                      {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z}, y0, y1, y2, tn))
                                        cold[kColdHost][lane] = 1.0f;  // the ray is the caller's (see bvh_trace.hip)
                                }
                            }
                        }
                    } else {
                        const float4 s3 = rec[3];
                        x2 = 0.0f;
                        if constexpr (ATTR != 0) {
                            // GeometricPrimitive::Intersect around a BilinearPatch: the recursion of cpu/primitive.cpp:
                            // 63-69 followed for up to three re-traces, as in bvh_trace.hip (ALPHA = 2)
                            constexpr int kAlphaPatchDepth = 3;
                            const float a = (flags & kPrimAlpha) ? s2.w : 1.0f;
                            RayState rn = r;
                            float tm = rayTMax, t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
                            int k = 0;
                            for (;;) {
                                hit = patch_test(rn, d, tm, {s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z},
                                                 {s3.x, s3.y, s3.z}, x0, x1, th);
                                if (!hit || !(a < 1)) break;
                                const float u = (a <= 0) ? 1.f : hash_float_6f(rn.o, d);
                                if (!(u > a)) break;
                                if (k == kAlphaPatchDepth) {
                                    hit = false;
                                    cold[kColdHost][lane] = 1.0f;
                                    break;
                                }
                                if (k == 0) t0 = th;
                                else if (k == 1) t1 = th;
                                else t2 = th;
                                ++k;
                                rn.o = patch_retrace_origin({s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z},
                                                            {s3.x, s3.y, s3.z}, x0, x1, (flags & kPrimFlipN) != 0, d,
                                                            (flags & kPrimSmooth) != 0, (flags & kPrimUV) != 0,
                                                            p.extras + 6 * (long)leafIdx, 4);
                                tm = tm - th;
                                tests += 1;
                            }
                            if (hit && k > 0) {
                                if (k == 3) th = th + t2;
                                if (k >= 2) th = th + t1;
                                th = th + t0;
                            }
                        } else {
                            hit = patch_test(r, d, rayTMax, {s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z},
                                             {s2.x, s2.y, s2.z}, {s3.x, s3.y, s3.z}, x0, x1, th);
                        }
                    }
                    if (hit) {
                        if (MODE == 0) {
                            cold[kColdHit][lane] = s0.w;
                            cold[kColdHit + 1][lane] = x0;
                            cold[kColdHit + 2][lane] = x1;
                            cold[kColdHit + 3][lane] = x2;
                            rayTMax = th;  // :1035-1036, :1046-1047
                        } else {
                            found = true;
                        }
                    }
                }
                leafLeft -= 1;
                leafIdx = nextIdx;
                leafPos += 1;
                if (MODE == 1 && found) cur = kKdDone;  // :1091-1094, :1101-1104
                else if (leafLeft == 0) pop_or_done();
            }
        } else {
            // ---- node step(s): up to p.nodeRepeat in a row before the next scheduling decision ------
            int rep = 0;
            do {
                if (cur >= 0) {
                    if (MODE == 0 && rayTMax < tMin) {  // :989-991 a hit closer than this node: finished
                        cur = kKdDone;
                    } else {
                        visited += 1;
                        const uint2 nd = *at(p.nodes, cur);
                        const unsigned flags = nd.y;
                        if ((flags & 3u) != 3u) {
                            // interior (:993-1023 / :1110-1144)
                            const int axis = (int)(flags & 3u);
                            const float split = __uint_as_float(nd.x);
                            const float oa = axis == 0 ? r.o.x : (axis == 1 ? r.o.y : r.o.z);
                            const float ia = axis == 0 ? r.inv.x : (axis == 1 ? r.inv.y : r.inv.z);
                            const bool dLe0 = ((r.kz >> (4 + axis)) & 1) != 0;  // ray.d[axis] <= 0
                            const float tSplit = (split - oa) * ia;
                            const bool belowFirst = (oa < split) || (oa == split && dLe0);
                            const int above = (int)(flags >> 2);
                            const int firstChild = belowFirst ? cur + 1 : above;
                            const int secondChild = belowFirst ? above : cur + 1;
                            if (tSplit > tMax || tSplit <= 0.0f) {
                                cur = firstChild;
                            } else if (tSplit < tMin) {
                                cur = secondChild;
                            } else {
                                if (sp - base == W) {  // window full: its oldest entry goes to HBM
                                    float4 e;
                                    e.x = stk[base & (W - 1)][0][lane];
                                    e.y = stk[base & (W - 1)][1][lane];
                                    e.z = stk[base & (W - 1)][2][lane];
                                    e.w = 0.0f;
                                    p.spill[(long)base * spillStride + gtid] = e;
                                    ++base;
                                }
                                stk[sp & (W - 1)][0][lane] = __int_as_float(secondChild);
                                stk[sp & (W - 1)][1][lane] = tSplit;
                                stk[sp & (W - 1)][2][lane] = tMax;
                                ++sp;
                                cur = firstChild;
                                tMax = tSplit;
                            }
                        } else {
                            // leaf (:1025-1061 / :1084-1107): one primitive index lives in the node itself,
                            // several sit in primitiveIndices — the first of them is fetched here, so that
                            // every primitive step is ONE trip to memory
                            const int nPrimitives = (int)(flags >> 2);
                            if (nPrimitives == 0) {
                                pop_or_done();
                            } else {
                                leafLeft = nPrimitives;
                                leafPos = (int)nd.x + 1;
                                leafIdx = nPrimitives == 1 ? (int)nd.x : *at(p.primIndices, (int)nd.x);
                                cur = kKdLeaf;
                            }
                        }
                    }
                }
            } while (++rep < p.nodeRepeat && __ballot(cur >= 0) != 0ull);
        }
    }
}

__global__ void kd_zero_queue_kernel(unsigned *queue, int words) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) queue[i] = 0u;
}

static bool kd_hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

struct KdDeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit KdDeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = kd_hip_ok(hipSetDevice(dev), "hipSetDevice");
    }
    ~KdDeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct KdWorkspace {
    unsigned *queue = nullptr;
    float4 *spill = nullptr;
    void *d_in = nullptr, *d_out = nullptr, *d_aux0 = nullptr, *d_aux1 = nullptr;
    size_t in_bytes = 0, out_bytes = 0, aux_bytes = 0;
};

static bool kd_grow(void **ptr, size_t *have, size_t need, const char *what) {
    if (*have >= need) return true;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    if (!kd_hip_ok(hipMalloc(ptr, need), what)) return false;
    *have = need;
    return true;
}

// shapes.cpp:176-177 with the reference's float32 operations (see bvh_capi.cpp)
static float kd_dop_host(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = std::fma(a, b, -cd);
    float err = std::fma(-c, d, cd);
    return diff + err;
}
static bool kd_triangle_is_degenerate(const float *p0, const float *p1, const float *p2) {
    float v[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    float w[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    float cx = kd_dop_host(v[1], w[2], v[2], w[1]);
    float cy = kd_dop_host(v[2], w[0], v[0], w[2]);
    float cz = kd_dop_host(v[0], w[1], v[1], w[0]);
    return cx * cx + cy * cy + cz * cz == 0.0f;
}

}  // namespace nnbvh

using namespace nnbvh;

struct nnbvh_kd_scene {
    int device = 0;
    int n_cus = 0;
    int depth = 0;
    int has_host_prims = 0;
    int has_patches = 0;
    int fits32 = 0;  // nodes (8 B), primitive records (64 B) and indices (4 B) each below 4 GiB
    float bounds[6];
    uint2 *d_nodes = nullptr;
    int32_t *d_indices = nullptr;
    float4 *d_prims = nullptr;
    float4 *d_extras = nullptr;  // 6 slots per primitive, scenes with attribute-reading alpha kinds only
    int blocks_per_cu[2] = {0, 0};
    std::mutex mu;
    std::map<hipStream_t, KdWorkspace> workspaces;
};

static KdWorkspace *kd_workspace_for(nnbvh_kd_scene *s, hipStream_t stream) {
    auto it = s->workspaces.find(stream);
    if (it != s->workspaces.end()) return &it->second;
    KdWorkspace w;
    // an entry is spilled only when W newer ones sit above it: levels 0 .. depth - W of a lane's list, at the
    // smallest window any instance runs with
    const size_t spill_levels = (size_t)std::max(s->depth + 1 - std::min(kKdW, kKdWLean), 1);
    const size_t spill_bytes = spill_levels * (size_t)s->n_cus * 8 * kKdBlock * sizeof(float4);
    if (!kd_hip_ok(hipMalloc((void **)&w.queue, kKdQueues * kKdQueueStride * sizeof(unsigned)), "hipMalloc(queue)"))
        return nullptr;
    if (!kd_hip_ok(hipMalloc((void **)&w.spill, spill_bytes), "hipMalloc(spill)")) {
        (void)hipFree(w.queue);
        return nullptr;
    }
    return &(s->workspaces[stream] = w);
}

static int kd_launch(nnbvh_kd_scene *s, int mode, const void *d_rays, int64_t n, void *d_hits, void *d_occ,
                     void *d_vis, void *d_tests, hipStream_t stream, KdWorkspace *w) {
    KdParams p;
    p.nodes = s->d_nodes;
    p.primIndices = s->d_indices;
    p.prims = s->d_prims;
    std::memcpy(p.bmin, s->bounds, 12);
    std::memcpy(p.bmax, s->bounds + 3, 12);
    p.rays = (const nnbvh_ray *)d_rays;
    p.hits = (nnbvh_hit *)d_hits;
    p.occluded = (uint8_t *)d_occ;
    p.visitedOut = (int32_t *)d_vis;
    p.testsOut = (int32_t *)d_tests;
    p.n = (long)n;
    p.queue = w->queue;
    p.nQueues = kKdQueues;
    p.primWeight = 12;  // tools/kd_sweep.sh on bathroom: 12 / 8 / 4 is the best of 4 x 2 x 3 (+1.5 % over 24 / 8 / 4)
    p.refillWeight = 8;
    p.nodeRepeat = 4;
#ifdef NNBVH_KD_TUNE  // tools only: scheduling knobs from the environment
    if (const char *e = getenv("NNBVH_KD_PRIMW")) p.primWeight = atoi(e);
    if (const char *e = getenv("NNBVH_KD_REFILLW")) p.refillWeight = atoi(e);
    if (const char *e = getenv("NNBVH_KD_NODEREP")) p.nodeRepeat = atoi(e);
#endif
    p.hasHostPrims = s->has_host_prims;
    p.spill = w->spill;
    // the four instances: closest / any hit x scenes with / without bilinear patches
    void (*const kernels[8])(KdParams) = {
        kd_trace_kernel<0, 0, kKdWLean, 0>, kd_trace_kernel<1, 0, kKdWLean, 0>, kd_trace_kernel<0, 1, kKdW, 0>,
        kd_trace_kernel<1, 1, kKdW, 0>,     kd_trace_kernel<0, 0, kKdWLean, 1>, kd_trace_kernel<1, 0, kKdWLean, 1>,
        kd_trace_kernel<0, 1, kKdW, 1>,     kd_trace_kernel<1, 1, kKdW, 1>};
    // ... and the attribute-reading forms of the PATCH instances
    void (*const attr_kernels[4])(KdParams) = {kd_trace_kernel<0, 1, kKdW, 0, 1>, kd_trace_kernel<1, 1, kKdW, 0, 1>,
                                               kd_trace_kernel<0, 1, kKdW, 1, 1>, kd_trace_kernel<1, 1, kKdW, 1, 1>};
    p.extras = s->d_extras;
    void (*const kernel)(KdParams) = s->d_extras ? attr_kernels[mode + 2 * s->fits32]
                                                 : kernels[mode + 2 * s->has_patches + 4 * s->fits32];
    if (s->blocks_per_cu[mode] == 0) {
        int occ = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kKdBlock, 0);
        s->blocks_per_cu[mode] = (e == hipSuccess && occ > 0) ? std::min(occ, 8) : 4;
    }
    int blocks = s->n_cus * s->blocks_per_cu[mode];
    const int64_t need = (n + kKdBlock - 1) / kKdBlock;
    if (need < blocks) blocks = (int)std::max<int64_t>(need, 1);
    hipLaunchKernelGGL(kd_zero_queue_kernel, dim3(1), dim3(256), 0, stream, w->queue, kKdQueues * kKdQueueStride);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kKdBlock), 0, stream, p);
    return kd_hip_ok(hipGetLastError(), "kd trace kernel launch") ? NNBVH_OK : NNBVH_ERR_DEVICE;
}

extern "C" {

nnbvh_kd_scene *nnbvh_kd_scene_create(const nnbvh_kd_node *nodes, int n_nodes, const int32_t *prim_indices,
                                      int n_indices, const nnbvh_prim *prims, int n_prims, const float *verts,
                                      int n_verts, const float bounds_min_max[6], int device) {
    return nnbvh_kd_scene_create_with_attributes(nodes, n_nodes, prim_indices, n_indices, prims, n_prims, verts, n_verts,
                                                 bounds_min_max, nullptr, nullptr, nullptr, device);
}

nnbvh_kd_scene *nnbvh_kd_scene_create_with_attributes(const nnbvh_kd_node *nodes, int n_nodes,
                                                      const int32_t *prim_indices, int n_indices,
                                                      const nnbvh_prim *prims, int n_prims, const float *verts,
                                                      int n_verts, const float bounds_min_max[6], const float *normals,
                                                      const float *uvs, const float *prim_alpha, int device) {
    if (!nodes || n_nodes <= 0 || n_indices < 0 || (n_indices > 0 && !prim_indices) || !prims || n_prims <= 0 ||
        !verts || n_verts <= 0 || !bounds_min_max) {
        set_error("kd_scene_create: null or empty argument");
        return nullptr;
    }
    // ---- validate everything the kernel indexes with ---------------------------------------------
    for (int i = 0; i < n_indices; ++i)
        if (prim_indices[i] < 0 || prim_indices[i] >= n_prims) {
            set_error("kd_scene_create: primitive index out of range in primitiveIndices");
            return nullptr;
        }
    struct Frame {
        int node, end, depth;
    };
    std::vector<Frame> st;
    st.push_back({0, n_nodes, 0});
    int visited = 0, max_depth = 0;
    while (!st.empty()) {
        const Frame f = st.back();
        st.pop_back();
        ++visited;
        if (f.node < 0 || f.node >= f.end) {
            set_error("kd_scene_create: node index outside its subtree range");
            return nullptr;
        }
        max_depth = std::max(max_depth, f.depth);
        const nnbvh_kd_node &nd = nodes[f.node];
        if ((nd.flags & 3u) == 3u) {
            if (f.node + 1 != f.end) {
                set_error("kd_scene_create: leaf does not close its subtree range (not the reference's layout)");
                return nullptr;
            }
            const uint32_t np = nd.flags >> 2;
            const int32_t v = (int32_t)nd.split_or_index;
            if (np == 1 && (v < 0 || v >= n_prims)) {
                set_error("kd_scene_create: leaf primitive index out of range");
                return nullptr;
            }
            if (np > 1 && (v < 0 || (int64_t)v + np > n_indices)) {
                set_error("kd_scene_create: leaf primitive range out of bounds");
                return nullptr;
            }
        } else {
            const int64_t above = nd.flags >> 2;
            if (above <= f.node + 1 || above >= f.end) {
                set_error("kd_scene_create: above-child link outside its subtree range");
                return nullptr;
            }
            float split;
            std::memcpy(&split, &nd.split_or_index, 4);
            if (std::isnan(split)) {
                set_error("kd_scene_create: NaN split position");
                return nullptr;
            }
            st.push_back({(int)above, f.end, f.depth + 1});
            st.push_back({f.node + 1, (int)above, f.depth + 1});
        }
    }
    if (visited != n_nodes) {
        set_error("kd_scene_create: unreachable nodes in the array");
        return nullptr;
    }
    if (max_depth > kMaxStack) {
        set_error("kd_scene_create: tree deeper than the traversal stack (64 entries, aggregates.cpp:982)");
        return nullptr;
    }
    // ---- primitive records: 4 slots per primitive in the caller's order --------------------------
    std::vector<float> rec((size_t)n_prims * 16, 0.0f);
    bool has_host = false, has_patch = false;
    // the alpha-tested kinds that read per-vertex attributes: on the device when the caller gave what they read
    // (6 more slots per primitive in a second array), else the host's as before
    auto on_device = [&](int kind) {
        if (is_smooth_alpha_kind(kind)) return normals != nullptr;
        if (is_alpha_patch_kind(kind))
            return prim_alpha && (!is_smooth_alpha_patch_kind(kind) || normals) && (!is_uv_alpha_patch_kind(kind) || uvs);
        return false;
    };
    bool any_attr = false;
    for (int k = 0; k < n_prims && !any_attr; ++k) any_attr = on_device(prims[k].kind);
    std::vector<float> extras(any_attr ? (size_t)n_prims * 24 : 0, 0.0f);
    for (int k = 0; k < n_prims; ++k) {
        const nnbvh_prim &pr = prims[k];
        float *s = &rec[(size_t)k * 16];
        uint32_t flags = 0;
        std::memcpy(&s[3], &pr.id, 4);
        const bool attr = on_device(pr.kind);
        if (pr.kind == NNBVH_PRIM_HOST || ((is_smooth_alpha_kind(pr.kind) || is_alpha_patch_kind(pr.kind)) && !attr)) {
            // (alpha-tested triangles of smooth meshes and alpha-tested patches without the arrays they read: the host's)
            flags |= kPrimHost;
            has_host = true;
        } else if (is_triangle_kind(pr.kind) || pr.kind == NNBVH_PRIM_BILINEAR_PATCH || is_alpha_patch_kind(pr.kind)) {
            const int nv = is_triangle_kind(pr.kind) ? 3 : 4;
            if (attr) {
                float *ex = &extras[(size_t)k * 24];
                const bool smooth = is_smooth_alpha_kind(pr.kind) || is_smooth_alpha_patch_kind(pr.kind);
                for (int j = 0; j < nv && smooth; ++j)
                    if (pr.v[j] >= 0 && pr.v[j] < n_verts) std::memcpy(&ex[4 * j], normals + 3 * (size_t)pr.v[j], 12);
                for (int j = 0; j < 4 && is_uv_alpha_patch_kind(pr.kind); ++j)
                    if (pr.v[j] >= 0 && pr.v[j] < n_verts) std::memcpy(&ex[16 + 2 * j], uvs + 2 * (size_t)pr.v[j], 8);
                flags |= kPrimAlpha | (smooth ? kPrimSmooth : 0u) | (is_uv_alpha_patch_kind(pr.kind) ? kPrimUV : 0u);
                if (pr.kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED || is_flipped_alpha_patch_kind(pr.kind)) flags |= kPrimFlipN;
                has_host = true;   // a re-trace chain that does not end voids the ray
                has_patch = true;  // the alpha test hashes the ray direction
            }
            if (pr.kind == NNBVH_PRIM_ALPHA_TRIANGLE || pr.kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED) {
                // the alpha value rides in v[3] (bit pattern) and goes to slot 2's w, as in the BVH scenes;
                // a re-trace that hits voids the ray like a host primitive, and the test needs the ray
                // direction: such scenes run the PATCH instances
                flags |= kPrimAlpha | (pr.kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED ? kPrimFlipN : 0u);
                has_host = true;
                has_patch = true;
            }
            for (int j = 0; j < nv; ++j) {
                if (pr.v[j] < 0 || pr.v[j] >= n_verts) {
                    set_error("kd_scene_create: vertex index out of range");
                    return nullptr;
                }
                std::memcpy(&s[4 * j], verts + 3 * (size_t)pr.v[j], 12);
            }
            std::memcpy(&s[3], &pr.id, 4);
            if (flags & kPrimAlpha) {
                if (is_alpha_patch_kind(pr.kind)) s[11] = prim_alpha[k];  // a patch needs all four v[]
                else std::memcpy(&s[11], &pr.v[3], 4);
            }
            if (nv == 4) {
                flags |= kPrimPatch;
                has_patch = true;
            }
            else if (kd_triangle_is_degenerate(&s[0], &s[4], &s[8])) flags |= kPrimDegenerate;
        } else {
            set_error("kd_scene_create: unsupported primitive kind (triangles, alpha-tested triangles, patches, host primitives)");
            return nullptr;
        }
        std::memcpy(&s[7], &flags, 4);
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        set_error("kd_scene_create: no usable HIP device");
        return nullptr;
    }
    KdDeviceGuard guard(device);
    if (!guard.ok) return nullptr;
    hipDeviceProp_t prop;
    if (!kd_hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return nullptr;
    auto *s = new nnbvh_kd_scene;
    s->device = device;
    s->n_cus = prop.multiProcessorCount;
    s->depth = max_depth;
    s->has_host_prims = has_host ? 1 : 0;
    s->has_patches = has_patch ? 1 : 0;
    s->fits32 = (n_nodes < (1 << 29) && n_prims < (1 << 26) - 1 && n_indices < (1 << 30)) ? 1 : 0;
    std::memcpy(s->bounds, bounds_min_max, 24);
    const size_t ni = (size_t)std::max(n_indices, 1);
    bool ok = kd_hip_ok(hipMalloc((void **)&s->d_nodes, (size_t)n_nodes * 8), "hipMalloc(kd nodes)") &&
              kd_hip_ok(hipMalloc((void **)&s->d_indices, ni * 4), "hipMalloc(kd indices)") &&
              kd_hip_ok(hipMalloc((void **)&s->d_prims, rec.size() * 4), "hipMalloc(kd prims)") &&
              kd_hip_ok(hipMemcpy(s->d_nodes, nodes, (size_t)n_nodes * 8, hipMemcpyHostToDevice), "upload kd nodes") &&
              kd_hip_ok(hipMemcpy(s->d_prims, rec.data(), rec.size() * 4, hipMemcpyHostToDevice), "upload kd prims");
    if (ok && n_indices > 0)
        ok = kd_hip_ok(hipMemcpy(s->d_indices, prim_indices, (size_t)n_indices * 4, hipMemcpyHostToDevice),
                       "upload kd indices");
    if (ok && any_attr)
        ok = kd_hip_ok(hipMalloc((void **)&s->d_extras, extras.size() * 4), "hipMalloc(kd attributes)") &&
             kd_hip_ok(hipMemcpy(s->d_extras, extras.data(), extras.size() * 4, hipMemcpyHostToDevice), "upload kd attributes");
    if (!ok) {
        nnbvh_kd_scene_destroy(s);
        return nullptr;
    }
    return s;
}

void nnbvh_kd_scene_destroy(nnbvh_kd_scene *s) {
    if (!s) return;
    KdDeviceGuard guard(s->device);
    for (auto &kv : s->workspaces) {
        KdWorkspace &w = kv.second;
        for (void *ptr : {(void *)w.queue, (void *)w.spill, w.d_in, w.d_out, w.d_aux0, w.d_aux1})
            if (ptr) (void)hipFree(ptr);
    }
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_indices) (void)hipFree(s->d_indices);
    if (s->d_prims) (void)hipFree(s->d_prims);
    if (s->d_extras) (void)hipFree(s->d_extras);
    delete s;
}

int nnbvh_kd_intersect_closest_device(nnbvh_kd_scene *s, const void *d_rays, int64_t n, void *d_hits,
                                      void *stream) {
    if (!s || n < 0 || n >= 0x7fffffffLL || (n > 0 && (!d_rays || !d_hits))) {
        set_error("kd_intersect_closest_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    KdDeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    KdWorkspace *w = kd_workspace_for(s, (hipStream_t)stream);
    if (!w) return NNBVH_ERR_DEVICE;
    return kd_launch(s, 0, d_rays, n, d_hits, nullptr, nullptr, nullptr, (hipStream_t)stream, w);
}

int nnbvh_kd_intersect_any_device(nnbvh_kd_scene *s, const void *d_rays, int64_t n, void *d_occluded,
                                  void *d_nodes_visited, void *d_prim_tests, void *stream) {
    if (!s || n < 0 || n >= 0x7fffffffLL || (n > 0 && (!d_rays || !d_occluded))) {
        set_error("kd_intersect_any_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    KdDeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    KdWorkspace *w = kd_workspace_for(s, (hipStream_t)stream);
    if (!w) return NNBVH_ERR_DEVICE;
    return kd_launch(s, 1, d_rays, n, nullptr, d_occluded, d_nodes_visited, d_prim_tests, (hipStream_t)stream, w);
}

int nnbvh_kd_intersect_closest(nnbvh_kd_scene *s, const nnbvh_ray *rays, int64_t n, nnbvh_hit *hits) {
    if (!s || n < 0 || n >= 0x7fffffffLL || (n > 0 && (!rays || !hits))) {
        set_error("kd_intersect_closest: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    KdDeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    KdWorkspace *w = kd_workspace_for(s, nullptr);
    if (!w) return NNBVH_ERR_DEVICE;
    if (!kd_grow(&w->d_in, &w->in_bytes, (size_t)n * sizeof(nnbvh_ray), "hipMalloc(rays)") ||
        !kd_grow(&w->d_out, &w->out_bytes, (size_t)n * sizeof(nnbvh_hit), "hipMalloc(hits)"))
        return NNBVH_ERR_DEVICE;
    if (!kd_hip_ok(hipMemcpy(w->d_in, rays, (size_t)n * sizeof(nnbvh_ray), hipMemcpyHostToDevice), "copy rays"))
        return NNBVH_ERR_DEVICE;
    const int rc = kd_launch(s, 0, w->d_in, n, w->d_out, nullptr, nullptr, nullptr, nullptr, w);
    if (rc != NNBVH_OK) return rc;
    if (!kd_hip_ok(hipStreamSynchronize(nullptr), "kd trace") ||
        !kd_hip_ok(hipMemcpy(hits, w->d_out, (size_t)n * sizeof(nnbvh_hit), hipMemcpyDeviceToHost), "copy hits"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_kd_intersect_any(nnbvh_kd_scene *s, const nnbvh_ray *rays, int64_t n, uint8_t *occluded,
                           int32_t *nodes_visited, int32_t *prim_tests) {
    if (!s || n < 0 || n >= 0x7fffffffLL || (n > 0 && (!rays || !occluded))) {
        set_error("kd_intersect_any: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    KdDeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    KdWorkspace *w = kd_workspace_for(s, nullptr);
    if (!w) return NNBVH_ERR_DEVICE;
    if (!kd_grow(&w->d_in, &w->in_bytes, (size_t)n * sizeof(nnbvh_ray), "hipMalloc(rays)") ||
        !kd_grow(&w->d_out, &w->out_bytes, (size_t)n, "hipMalloc(occluded)") ||
        !kd_grow(&w->d_aux0, &w->aux_bytes, (size_t)n * 8, "hipMalloc(counts)"))
        return NNBVH_ERR_DEVICE;
    int32_t *d_vis = (int32_t *)w->d_aux0, *d_tst = d_vis + n;
    if (!kd_hip_ok(hipMemcpy(w->d_in, rays, (size_t)n * sizeof(nnbvh_ray), hipMemcpyHostToDevice), "copy rays"))
        return NNBVH_ERR_DEVICE;
    const int rc = kd_launch(s, 1, w->d_in, n, nullptr, w->d_out, nodes_visited ? d_vis : nullptr,
                             prim_tests ? d_tst : nullptr, nullptr, w);
    if (rc != NNBVH_OK) return rc;
    if (!kd_hip_ok(hipStreamSynchronize(nullptr), "kd trace") ||
        !kd_hip_ok(hipMemcpy(occluded, w->d_out, (size_t)n, hipMemcpyDeviceToHost), "copy occluded"))
        return NNBVH_ERR_DEVICE;
    if (nodes_visited && !kd_hip_ok(hipMemcpy(nodes_visited, d_vis, (size_t)n * 4, hipMemcpyDeviceToHost), "copy counts"))
        return NNBVH_ERR_DEVICE;
    if (prim_tests && !kd_hip_ok(hipMemcpy(prim_tests, d_tst, (size_t)n * 4, hipMemcpyDeviceToHost), "copy counts"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

}  // extern "C"
