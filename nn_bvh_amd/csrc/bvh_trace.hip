// bvh_trace.hip — hand-written gfx950 traversal kernels for the nnbvh C ABI.
//
// What is computed (reference: /root/reference/src/pbrt):
//   closest hit   BVHAggregate::Intersect   cpu/aggregates.cpp:529-579
//   any hit       BVHAggregate::IntersectP  cpu/aggregates.cpp:581-624
//   slab test     Bounds3::IntersectP       util/vecmath.h:1573-1608
//   triangle      IntersectTriangle         shapes.cpp:172-273
//   patch         IntersectBilinearPatch    shapes.h:1279-1347 (+ util/math.h:614-637, 1420-1426)
// Results (hit primitive, t, barycentrics, node-visit and primitive-test counts) are
// bit-identical to that code; HOW it is computed is CDNA4-specific:
//   * persistent 64-lane wavefronts pull rays from a global queue; lanes whose ray has
//     terminated are refilled in place (ballot + mbcnt compaction of the idle lanes), so
//     SIMD slots stay occupied although rays visit 1..400 nodes each;
//   * one 64-B "both children" record per interior node (nnbvh_internal.h) halves the
//     length of the dependent-load chain of the reference's 32-B node walk; a child box's
//     whole slab test is carried by ONE float, its entry distance or +inf (trace_math.h
//     slab_entry_key), and the far child is pushed WITH it, so the deferred box test the
//     reference performs when it pops the node (`tMin < tMax` with the then-current tMax)
//     needs no memory access at all and still gives the identical boolean;
//   * the per-lane traversal stack is a short ring window in LDS ([entry][word][lane], bank-
//     conflict-free) that spills its oldest entry to a coalesced HBM scratch array only
//     when a ray's pending-node list outgrows the window;
//   * fp32 arithmetic is emitted with -ffp-contract=off; __builtin_fmaf appears exactly
//     where the reference calls FMA(); division and sqrt are IEEE (hipcc default).
// No MFMA: this is pointer chasing + branchy fp32, bounded by cache/HBM request rate.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_trace.h"
#include "anim_math.h"
#include "spawn_math.h"
#include "trace_math.h"

namespace nnbvh {

// ------------------------------------------------------------------------------------
constexpr int kDone = (int)0x80000000;    // never a leaf ref: ~slot with slot = 0x7fffffff
constexpr int kReturn = (int)0x80000001;  // an instance's child traversal is exhausted (slot 0x7ffffffe)
constexpr int kEnter = (int)0x80000002;   // INST = 2: the lane waits to enter an AnimatedPrimitive (slot 0x7ffffffd)

// MODE 0: closest hit (counts always)
// MODE 1: any hit with exact node-visit / prim-test counts (pushes every far child)
// MODE 2: any hit, occlusion flag only
// MODE 3: several batches in ONE launch, each closest hit (as MODE 0) or occlusion-only any hit (as
//         MODE 2): the batches of one wavefront iteration share a single ramp-up and a single drain
//         (DESIGN.md §5.1).  A lane's ray tag carries its batch; what differs per lane is only what a
//         hit does and what is written at retire.
//
// Every lane is a small state machine over `cur`:
//     cur >= 0            an interior record to process        (interior step)
//     cur <  0, != kDone  ~cur = prim-stream slot to test next (primitive step)
//     cur == kDone        no ray (retire the finished one, fetch the next)
// Each trip of the scheduling loop the WAVE picks one kind of step (wave-uniform, from
// ballots) and the lanes in that state execute it; the others idle for that trip.  A ray's
// own sequence of steps is exactly the reference's, so results cannot depend on the policy.
//
// waves/SIMD the register allocator must leave room for (measured in steady state: closest
// hit 5 -> 6 waves is +9 %; beyond that the spills cost more than the extra waves hide)
#ifndef NNBVH_MINW_CLOSEST
#define NNBVH_MINW_CLOSEST 6
#endif
#ifndef NNBVH_MINW_ANY
#define NNBVH_MINW_ANY 6
#endif
// the ALPHA instances (hash + re-trace in the primitive step): at 6 waves they spill 8-18 registers, at 5 none —
// a 1 M-triangle soup with 70 % alpha-tested triangles: closest 6.26 -> 4.95 ms, any 5.38 -> 4.52 ms
#ifndef NNBVH_MINW_ALPHA
#define NNBVH_MINW_ALPHA 5
#endif
// INST = 2: weight of a lane waiting to enter an AnimatedPrimitive in the step selection (interior = 16); 0 = enter
// at once inside the primitive step (the round-2 form: 11 of 64 lanes active in the interpolation)
// ALPHA = 2 (alpha-tested bilinear patches: the patch's interaction point and normal incl. the (s, t)
// reparametrisation, the re-trace loop): 158-162 VGPRs = 3 waves without spills; 2 waves measured 30 % slower,
// 4 waves (22-27 spilled registers) no faster
#ifndef NNBVH_MINW_ALPHA_PATCH
#define NNBVH_MINW_ALPHA_PATCH 3
#endif
#ifndef NNBVH_ANIM_ENTER_WEIGHT
#define NNBVH_ANIM_ENTER_WEIGHT 8
#endif
#ifndef NNBVH_FUSED_PRIM_LOOP
#define NNBVH_FUSED_PRIM_LOOP 0
#endif
// 1: the lean instances keep the ray-invariant select predicates (direction signs, kz) as wave-level masks
// in SGPRs (trace_math.h RayMasks) instead of recomputing them per lane in every step
#ifndef NNBVH_MASKS
#define NNBVH_MASKS 1
#endif
#ifndef NNBVH_LEAN_PRIM_LOOP
#define NNBVH_LEAN_PRIM_LOOP 1
#endif
#ifndef NNBVH_LEAN_EXTRA_WAVES
#define NNBVH_LEAN_EXTRA_WAVES 2
#endif
// 1: the lean instances run MERGED trips (see the scheduling loop): every lane with a node OR a primitive
// pending fetches in the same trip, through one load sequence, and the two kinds of arithmetic follow
// each other — a lane waiting on a leaf no longer sits out the interior trips of its wavefront.
#ifndef NNBVH_MERGED
#define NNBVH_MERGED 0
#endif
#ifndef NNBVH_FAT
#define NNBVH_FAT 0
#endif
// 1: a far child pushed and popped within the same step is taken from registers (experiment)
#ifndef NNBVH_FWD_TOP
#define NNBVH_FWD_TOP 0
#endif
// SOA = 1 (lean instances of modes 0, 2, 3 only): a batch without nnbvh_ray records (rays == nullptr) is read as the
// SOA<Ray> slices of a wavefront queue — no gather pass.  Its own instances: the mere presence of the second fetch
// path in the refill trip cost the one-launch step 1.3 % (9.31 -> 9.44 ms).
//
// INST = 1: the scene is two-level (TransformedPrimitive leaves, cpu/primitive.cpp:112-131).  An
// instance primitive saves the lane's ray state in LDS, transforms the ray with the reference's
// interval arithmetic, traverses the child tree ABOVE the current stack level (`floor`) and
// returns to the outer leaf when the child is exhausted.  Compiled separately so that
// single-level scenes pay nothing for it.
//
// INST = 2: ... with AnimatedPrimitives among the instances (the interpolation of the transform costs 60
// VGPRs; scenes whose instances are all static run INST = 1).
//
// PATCH = 0 ("lean"): the scene holds no bilinear patches, no instances and no host-only primitives, so
// nothing ever reads the ray direction after the ray is fetched: no patch test, no direction / ray
// index / host flag in LDS (20 KiB per block = 8 blocks per CU), <= 64 VGPRs = 8 wavefronts per SIMD,
// records and slots through 32-bit offsets from a scalar base, one primitive step per decision.
// ALPHA = 1: the scene holds alpha-tested triangles (kPrimAlpha, cpu/primitive.cpp:57-70); compiled
// separately so that other scenes pay nothing for the hash and the re-trace.
template <int MODE, int W, int INST, int PATCH, int ALPHA = 0, int SOA = 0>
__global__ __launch_bounds__(kBlockThreads, (INST ? 1 : (ALPHA ? (ALPHA == 2 ? NNBVH_MINW_ALPHA_PATCH : NNBVH_MINW_ALPHA) : ((MODE == 0 || MODE == 3) ? NNBVH_MINW_CLOSEST : NNBVH_MINW_ANY) + (PATCH ? 0 : NNBVH_LEAN_EXTRA_WAVES))))
void trace_kernel(TraceParams p) {
    static_assert(PATCH || !INST, "two-level scenes need the ray direction");
    static_assert(PATCH || !ALPHA, "the alpha test hashes the ray direction");
    // the stack window: entry k of a lane = (child reference, entry distance), the two words 64 dwords
    // apart so that one ds_read2st64 / ds_write2st64 with one address moves both
    __shared__ float s_stack[kBlockThreads / 64][W][2][64];
    // Cold per-ray state parked in LDS ([field][lane], conflict-free) instead of VGPRs: the
    // ray index (read when the ray retires), the direction (read by the patch test only; the
    // triangle test uses the precomputed shear) and, closest hit, the current best hit
    // (written on an accepted hit, read at retire).  Frees 4 / 8 registers per lane.
    // The lean instances (PATCH = 0: no patches, no instances, and — the launcher sees to it — no
    // host-only primitives) keep the ray index in a VGPR and have no host flag: the closest-hit kernel's
    // cold state is then the four hit words, 20 KiB of LDS per block with the stack window, which is
    // what lets an EIGHTH block share the CU's 160 KiB (its 61 VGPRs allow 8 wavefronts per SIMD).
    constexpr bool kLean = !PATCH && !INST;
    constexpr int kColdRi = 0, kColdD = 1, kColdHit = kLean ? 0 : (PATCH ? 4 : 1),
                  kColdHost = kColdHit + ((MODE == 0 || MODE == 3) ? 4 : 0);
    constexpr int kColdBase = kColdHost + (kLean ? 0 : 1);  // kColdHost: the ray reached a host-only primitive
    // two-level scenes: the outer ray saved while a child tree is traversed
    constexpr int kSaveO = kColdBase, kSaveInv = kColdBase + 3, kSaveShear = kColdBase + 6,
                  kSaveKz = kColdBase + 9, kSaveTmax = kColdBase + 10, kSaveD = kColdBase + 11,
                  kResume = kColdBase + 14, kCurInst = kColdBase + 15, kHitInst = kColdBase + 16,
                  kInnerHit = kColdBase + 17, kColdTime = kColdBase + 18, kPendSlot = kColdBase + 19;
    constexpr int kColdFields = kColdBase + (INST ? 20 : 0);
    __shared__ float s_cold[kBlockThreads / 64][kColdFields > 0 ? kColdFields : 1][64];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int gtid = blockIdx.x * kBlockThreads + threadIdx.x;
    float(*stk)[2][64] = s_stack[wave];
    float(*cold)[64] = s_cold[wave];
    int riReg = -1;  // lean instances: the ray this lane carries, -1 = none
    if (!kLean) cold[kColdRi][lane] = __int_as_float(-1);
    const long spillStride = (long)gridDim.x * kBlockThreads;

    // which queue this wave drains first: its XCD's share of the batch (speed only)
    int q = 0;
    if (p.nQueues > 1) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        q = (int)(xcc & 0xf) % p.nQueues;
    }
    int queuesTried = 0;
    int curBatch = 0;  // MODE 3: the batch this wave is drawing rays from (wave-uniform)
    // batch size: the host's bound, or a device-resident queue size below it (wavefront callers)
    long nRays = p.n;
    if (p.nDev) {
        const long nd = (long)*p.nDev;
        nRays = nd < 0 ? 0 : (nd < nRays ? nd : nRays);
    }

#ifdef NNBVH_STATS
    // trips/lanes per kind (I, P, R); [6..8] = sum nInt,nPrim,nIdle over I trips; [10..12] = shader
    // cycles (s_memtime) spent in I / P / R trips incl. their scheduling decision; [14], [15] =
    // interior steps executed and the lanes that took part in them
    unsigned long long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stPrev = __builtin_amdgcn_s_memtime();
    int stKind = 2;
#endif
    RayState r;
    RayMasks masks = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};  // lean instances: rebuilt after every refill trip
    float tMax = 0.0f;
    int visited = 0, tests = 0;
    int cur = kDone, sp = 0, base = 0;
    int floor = -1;      // INST: stack level the current child traversal must not pop below
    bool found = false;  // MODE 1/2
    bool exhausted = false;

    // pop entries until one whose deferred box test passes with the current tMax
    auto pop_entry = [&](int &ref, float &key) {
        --sp;
        ref = __float_as_int(stk[sp & (W - 1)][0][lane]);  // always an LDS read (stale if spilled)
        key = stk[sp & (W - 1)][1][lane];
        asm volatile("" : "+v"(ref), "+v"(key));  // keeps the two reads ds_read (no flat select)
        if (sp < base) {                     // rare: the entry lives in the HBM spill array
            const uint2 e = p.spill[(long)sp * spillStride + gtid];
            base = sp;
            ref = (int)e.x;
            key = __uint_as_float(e.y);
            // complete the load inside this rare branch, so that the common path's join
            // needs no vmcnt wait
            asm volatile("" : "+v"(ref), "+v"(key));
        }
        if (MODE != 2) visited += 1;
    };
    auto pop_next = [&]() -> int {
        const int lowest = (INST && floor > 0) ? floor : 0;
        const int none = (INST && floor >= 0) ? kReturn : kDone;
        if (sp <= lowest) return none;
        // the first entry is usually the one (only a tMax that shrank since the push rejects it)
        int ref;
        float key;
        pop_entry(ref, key);
        if (key < tMax) return ref;
        while (sp > lowest) {
            pop_entry(ref, key);
            if (key < tMax) return ref;
        }
        return none;
    };

    // TransformedPrimitive::Intersect / IntersectP (cpu/primitive.cpp:112-131): park the outer ray
    // in LDS, transform it into the instance's space, count and test the child root.
    auto enter_instance = [&](int slot, unsigned flags, float4 s0, float4 s1, float4 s2) {
        const float4 s3 = p.prims[slot + 3], s4 = p.prims[slot + 4], s5 = p.prims[slot + 5];
        const V3 dOuter = {cold[kColdD][lane], cold[kColdD + 1][lane], cold[kColdD + 2][lane]};
        cold[kSaveO][lane] = r.o.x;
        cold[kSaveO + 1][lane] = r.o.y;
        cold[kSaveO + 2][lane] = r.o.z;
        cold[kSaveInv][lane] = r.inv.x;
        cold[kSaveInv + 1][lane] = r.inv.y;
        cold[kSaveInv + 2][lane] = r.inv.z;
        cold[kSaveShear][lane] = r.sx;
        cold[kSaveShear + 1][lane] = r.sy;
        cold[kSaveShear + 2][lane] = r.sz;
        cold[kSaveKz][lane] = __int_as_float(r.kz);
        cold[kSaveTmax][lane] = tMax;
        cold[kSaveD][lane] = dOuter.x;
        cold[kSaveD + 1][lane] = dOuter.y;
        cold[kSaveD + 2][lane] = dOuter.z;
        cold[kResume][lane] = __int_as_float((flags & kPrimLast) ? kDone : ~(slot + 6));
        cold[kCurInst][lane] = __int_as_float(__float_as_int(s0.w) + 1);
        cold[kInnerHit][lane] = 0.0f;
        V3 oIn, dIn;
        float4 m0 = s2, m1 = s3, m2 = s4;
        if (INST == 2 && (flags & kPrimAnimated) && p.anim)  // AnimatedPrimitive: Interpolate(r.time), primitive.cpp:143-144
            anim_inverse_rows(p.anim + (long)kAnimStride * __float_as_int(s0.w), cold[kColdTime][lane], m0, m1, m2);
        apply_inverse_ray(m0, m1, m2, r.o, dOuter, tMax, oIn, dIn);
        r.o = oIn;
        cold[kColdD][lane] = dIn.x;
        cold[kColdD + 1][lane] = dIn.y;
        cold[kColdD + 2][lane] = dIn.z;
        r.inv = {1.0f / dIn.x, 1.0f / dIn.y, 1.0f / dIn.z};
        ray_shear(r, dIn);
        floor = sp;
        if (MODE != 2) visited += 1;  // the child aggregate's root
        float tEntry;
        const bool rootHit =
            slab_partial(s0.x, s0.y, s0.z, s1.x, s1.y, s1.z, r, tEntry) && (tEntry < tMax);
        cur = rootHit ? __float_as_int(s5.x) : kReturn;
    };

    // the child tree is exhausted: restore the outer ray; the inner tHit becomes the outer tMax
    // iff a hit was accepted inside (`si = primSi; tMax = si->tHit`), then resume the outer leaf
    auto leave_instance = [&]() {
        const bool innerHit = cold[kInnerHit][lane] != 0.0f;
        r.o = {cold[kSaveO][lane], cold[kSaveO + 1][lane], cold[kSaveO + 2][lane]};
        r.inv = {cold[kSaveInv][lane], cold[kSaveInv + 1][lane], cold[kSaveInv + 2][lane]};
        r.sx = cold[kSaveShear][lane];
        r.sy = cold[kSaveShear + 1][lane];
        r.sz = cold[kSaveShear + 2][lane];
        r.kz = __float_as_int(cold[kSaveKz][lane]);
        cold[kColdD][lane] = cold[kSaveD][lane];
        cold[kColdD + 1][lane] = cold[kSaveD + 1][lane];
        cold[kColdD + 2][lane] = cold[kSaveD + 2][lane];
        if (!innerHit) tMax = cold[kSaveTmax][lane];
        cold[kCurInst][lane] = 0.0f;
        floor = -1;
        const int resume = __float_as_int(cold[kResume][lane]);
        cur = (resume == kDone) ? pop_next() : resume;
    };

    // one primitive of the lane's leaf, its three slots s0..s2 in hand: the Primitive tag dispatch
    // (cpu/primitive.h), the leaf test, what a hit does, and where the lane goes next
    auto prim_math = [&](const int slot, const float4 s0, const float4 s1, const float4 s2) {
        const unsigned flags = __float_as_uint(s1.w);
        if (INST && (flags & kPrimInstance)) {
            if (INST == 2 && NNBVH_ANIM_ENTER_WEIGHT > 0 && (flags & kPrimAnimated) && p.anim) {
                // Interpolate(ray.time) costs ~600 instructions: the lane waits in the kEnter state until the
                // wave runs an ENTER step for all lanes that have reached an AnimatedPrimitive by then
                cold[kPendSlot][lane] = __int_as_float(slot);
                cur = kEnter;
            } else {
                enter_instance(slot, flags, s0, s1, s2);
            }
        } else if (!kLean && (flags & kPrimHost)) {
            // a primitive only the host can intersect (quadric, curve, alpha-tested
            // ...): this ray's result is void and the caller re-traces it on the CPU
            cold[kColdHost][lane] = 1.0f;
            cur = (flags & kPrimLast) ? pop_next() : ~(slot + 3);
        } else {
            tests += 1;
            bool hit;
            float x0, x1, x2, th;
            int next;
            if (!PATCH || !(flags & kPrimPatch)) {
                hit = triangle_test(r, tMax, (flags & kPrimDegenerate) != 0,
                                    {s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z},
                                    {s2.x, s2.y, s2.z}, x0, x1, x2, th, (kLean && NNBVH_MASKS) ? &masks : nullptr);
                next = slot + ((ALPHA && (flags & kPrimSmooth)) ? 6 : 3);
                if (ALPHA && hit && (flags & kPrimAlpha)) {
                    // GeometricPrimitive::Intersect, cpu/primitive.cpp:57-70 (IntersectP takes
                    // the same route, :79-81): stochastic alpha test on the ray as given
                    const float a = s2.w;
                    if (a < 1) {
                        const V3 rd = {cold[kColdD][lane], cold[kColdD + 1][lane],
                                       cold[kColdD + 2][lane]};
                        const float u = (a <= 0) ? 1.f : hash_float_6f(r.o, rd);
                        if (u > a) {
                            // ignored; the reference re-traces from the hit point against this
                            // shape alone: rNext = si->intr.SpawnRay(r.d), Intersect(rNext, tMax - tHit)
                            hit = false;
                            RayState rn = r;  // same direction: same reciprocals and shear
                            if (flags & kPrimSmooth) {  // a mesh with shading normals: offset along FaceForward(n, ns)
                                const float4 m0 = p.prims[slot + 3], m1 = p.prims[slot + 4], m2 = p.prims[slot + 5];
                                rn.o = alpha_retrace_origin({s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z}, x0,
                                                            x1, x2, (flags & kPrimFlipN) != 0, rd, true, {m0.x, m0.y, m0.z},
                                                            {m1.x, m1.y, m1.z}, {m2.x, m2.y, m2.z});
                            } else {
                                rn.o = alpha_retrace_origin({s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z},
                                                            {s2.x, s2.y, s2.z}, x0, x1, x2,
                                                            (flags & kPrimFlipN) != 0, rd);
                            }
                            tests += 1;  // Triangle::Intersect counts the re-test too
                            float y0, y1, y2, tn;
                            if (triangle_test(rn, tMax - th, (flags & kPrimDegenerate) != 0,
                                              {s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z},
                                              {s2.x, s2.y, s2.z}, y0, y1, y2, tn))
                                cold[kColdHost][lane] = 1.0f;  // never for a planar triangle; if
                                                               // it happens the ray is the caller's
                        }
                    }
                }
            } else {
                const float4 s3 = p.prims[slot + 3];
                x2 = 0.0f;
                const V3 rd = {cold[PATCH ? kColdD : 0][lane], cold[PATCH ? kColdD + 1 : 0][lane],
                               cold[PATCH ? kColdD + 2 : 0][lane]};
                next = slot + 4 + ((ALPHA == 2 && (flags & kPrimSmooth)) ? 4 : 0) + ((ALPHA == 2 && (flags & kPrimUV)) ? 2 : 0);
                if constexpr (ALPHA == 2) {
                    // GeometricPrimitive::Intersect around a BilinearPatch (cpu/primitive.cpp:50-70).  A non-planar
                    // patch can be met again by the ray spawned off its own surface, so the recursion of :63-69 is
                    // followed: level k tests the ray of level k (its origin goes into the hash), up to
                    // kAlphaPatchDepth re-traces, beyond which the record is void (the caller re-traces the ray)
                    constexpr int kAlphaPatchDepth = 3;
                    const float a = (flags & kPrimAlpha) ? s2.w : 1.0f;
                    RayState rn = r;  // patch_test reads the origin only
                    float tm = tMax, t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
                    int k = 0;
                    for (;;) {
                        hit = patch_test(rn, rd, tm, {s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z},
                                         {s3.x, s3.y, s3.z}, x0, x1, th);
                        if (!hit || !(a < 1)) break;  // :52-54 / :58
                        const float u = (a <= 0) ? 1.f : hash_float_6f(rn.o, rd);
                        if (!(u > a)) break;  // accepted: (x0, x1, th) are this level's
                        if (k == kAlphaPatchDepth) {
                            hit = false;
                            cold[kColdHost][lane] = 1.0f;
                            break;
                        }
                        if (k == 0) t0 = th;
                        else if (k == 1) t1 = th;
                        else t2 = th;
                        ++k;
                        // rNext = si->intr.SpawnRay(r.d); Intersect(rNext, tMax - si->tHit)
                        rn.o = patch_retrace_origin({s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z}, {s2.x, s2.y, s2.z},
                                                    {s3.x, s3.y, s3.z}, x0, x1, (flags & kPrimFlipN) != 0, rd,
                                                    (flags & kPrimSmooth) != 0, (flags & kPrimUV) != 0,
                                                    p.prims + slot + 4);
                        tm = tm - th;
                        tests += 1;
                    }
                    if (hit && k > 0) {  // siNext->tHit += si->tHit (:67-68), unwinding from the deepest level
                        if (k == 3) th = th + t2;
                        if (k >= 2) th = th + t1;
                        th = th + t0;
                    }
                } else {
                    hit = patch_test(r, rd, tMax, {s0.x, s0.y, s0.z}, {s1.x, s1.y, s1.z},
                                     {s2.x, s2.y, s2.z}, {s3.x, s3.y, s3.z}, x0, x1, th);
                }
            }
            bool closestLane = MODE == 0;
            if (MODE == 3 && hit)
                closestLane = !((p.anyMask >> ((kLean ? riReg : __float_as_int(cold[kColdRi][lane])) >> kFusedIndexBits)) & 1u);
            if (hit) {
                if (closestLane) {
                    cold[kColdHit][lane] = s0.w;  // primitive id bits
                    cold[kColdHit + 1][lane] = x0;
                    cold[kColdHit + 2][lane] = x1;
                    cold[kColdHit + 3][lane] = x2;
                    tMax = th;
                    if (INST) {
                        cold[kHitInst][lane] = cold[kCurInst][lane];
                        cold[kInnerHit][lane] = 1.0f;
                    }
                } else {
                    found = true;
                }
            }
            if (MODE != 0 && found) cur = kDone;           // aggregates.cpp:597-602
            else if (flags & kPrimLast) cur = pop_next();  // leaf finished
            else cur = ~next;
        }
    };

    // one interior record q0..q3 in hand: both children's slab keys, the far child pushed with its key,
    // the near child entered (aggregates.cpp:556-574)
    auto interior_math = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3) {
        const int ref0 = __float_as_int(q3.x), ref1 = __float_as_int(q3.y);
        const int axis = __float_as_int(q3.z);
        // aggregates.cpp:562-568: near child = second child iff dirIsNeg[axis]
        const bool swap = ((r.kz >> axis) & 1) != 0;  // dirIsNeg[axis], packed by ray_shear
        // one float per child: its entry distance, +inf if the box is missed whatever tMax is
        // (slab_entry_key) — the verdicts are then two compares against tMax
        const float k0 = (kLean && NNBVH_MASKS) ? slab_entry_key(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, masks)
                                                : slab_entry_key(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r);
        const float k1 = (kLean && NNBVH_MASKS) ? slab_entry_key(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r, masks)
                                                : slab_entry_key(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r);
        const int nearRef = swap ? ref1 : ref0, farRef = swap ? ref0 : ref1;
        const float nearT = swap ? k1 : k0, farT = swap ? k0 : k1;
        visited += 1;  // the near child is entered now
        // A far child whose tMax-independent tests failed (key +inf) can never be entered: it is
        // counted now and not pushed.  One that merely fails against TODAY's tMax must be pushed:
        // tMax is not monotone — a hit accepted with tScaled <= tMax * det can round to a t one ulp
        // ABOVE the old tMax (shapes.cpp:239-244; rays that meet a shared vertex at exactly tMax do
        // it), and the reference tests the far child against that later value.  MODE 1 pushes every
        // far child (exact counts up to the first hit).
        const bool doPush = (MODE == 1) || (farT < __builtin_inff());
#if NNBVH_FWD_TOP
        // near child missed, far child hit: the entry about to be pushed is the one pop_next would hand straight back —
        // take it from the registers (no LDS write + read, no spill of the oldest entry on its behalf)
        const bool takeFar = doPush && !(nearT < tMax) && (farT < tMax);
        if (!takeFar) {
#else
        constexpr bool takeFar = false;
        {
#endif
            if (doPush && sp - base == W - 1) {
                uint2 e;
                e.x = __float_as_uint(stk[base & (W - 1)][0][lane]);
                e.y = __float_as_uint(stk[base & (W - 1)][1][lane]);
                p.spill[(long)base * spillStride + gtid] = e;
                ++base;
            }
            stk[sp & (W - 1)][0][lane] = __int_as_float(farRef);
            stk[sp & (W - 1)][1][lane] = farT;
            sp += doPush ? 1 : 0;
        }
        if (MODE == 0 || MODE == 3) visited += doPush ? 0 : 1;
        if (nearT < tMax) cur = nearRef;
        else if (takeFar) {
            if (MODE != 2) visited += 1;  // as pop_entry counts the popped entry
            cur = farRef;
        } else cur = pop_next();
    };
    auto interior_step = [&]() {
#ifdef NNBVH_STATS
        st[14] += 1;
        st[15] += __popcll(__ballot(cur >= 0));
#endif
        if (cur >= 0) {
#ifdef NNBVH_PROBE_SALU  // sensitivity probes (tools only): extra scalar / vector instructions per interior step
#pragma unroll
            for (int k = 0; k < NNBVH_PROBE_SALU; ++k) asm volatile("s_add_u32 s95, s95, 1" ::: "s95", "scc");
#endif
#ifdef NNBVH_PROBE_VALU
#pragma unroll
            for (int k = 0; k < NNBVH_PROBE_VALU; ++k) asm volatile("v_add_u32 %0, %0, 1" : "+v"(tests));
#endif
            // lean instances address records and slots with a 32-bit byte offset from a scalar base (one
            // 32-bit shift instead of a 64-bit shift and a 64-bit add per fetch); the launcher only picks
            // them when both arrays are below 4 GiB (p.fits32)
            if constexpr (NNBVH_FAT && kLean) {
            // experiment (bvh_layout.cpp mode 32): 192-B records = the node's own record + copies of both
            // children's; the copy of the child this ray enters next arrives with the node's record, and the
            // lanes that do enter it (and find it interior) take that step in the same trip
            const unsigned at = ((unsigned)cur >> 2) * 192u;
            const unsigned nearCopy = 64u + (((unsigned)r.kz >> (cur & 3)) & 1u) * 64u;
            const float4 *rec = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.wide) + at);
            const float4 *rec2 = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.wide) + at + nearCopy);
            const float4 q3 = rec[3];
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            const float4 n3 = rec2[3];
            const float4 n0 = rec2[0], n1 = rec2[1], n2 = rec2[2];
            const int nearChild = (((unsigned)r.kz >> (cur & 3)) & 1u) ? __float_as_int(q3.y) : __float_as_int(q3.x);
            interior_math(q0, q1, q2, q3);
            // the near child was entered iff the lane now stands on it; a popped entry with the same
            // reference cannot exist (a node has one parent and is pushed at most once)
            if (cur >= 0 && cur == nearChild) interior_math(n0, n1, n2, n3);
            } else {
            const float4 *rec = kLean ? reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.wide) + ((unsigned)cur << 6))
                                      : p.wide + 4 * (long)cur;
            const float4 q3 = rec[3];
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            interior_math(q0, q1, q2, q3);
            }
        }
    };

    for (;;) {
#ifdef NNBVH_STATS
        {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            st[10 + stKind] += now - stPrev;
            stPrev = now;
        }
#endif
        const bool isInt = cur >= 0;
        const bool isIdle = cur == kDone;
        const int nInt = __popcll(__ballot(isInt));
        const unsigned long long idleMask = __ballot(isIdle);
        const int nIdle = __popcll(idleMask);
        const int nEnter = (INST == 2 && NNBVH_ANIM_ENTER_WEIGHT > 0) ? __popcll(__ballot(cur == kEnter)) : 0;
        const int nPrim = 64 - nInt - nIdle - nEnter;

        // Wave-uniform choice of this trip's step kind: the one that advances the most lanes
        // per instruction issued.  A lane's weight says how cheap its step is relative to an
        // interior step (weight 16): primitive tests and refills are longer, so they must
        // gather more lanes before they are worth a trip.  All-idle always refills/retires.
        const int sI = nInt * 16, sP = nPrim * p.primWeight;
        const int sR = exhausted ? 0 : nIdle * p.refillWeight;
#ifdef NNBVH_STATS
        unsigned long long &stTrips = (nIdle == 64 || (sR > sI && sR > sP)) ? st[4]
                                      : ((sP > sI || nInt == 0) ? st[2] : st[0]);
        (&stTrips)[0] += 1;
        (&stTrips)[1] += (nIdle == 64 || (sR > sI && sR > sP)) ? nIdle
                         : ((sP > sI || nInt == 0) ? nPrim : nInt);
        stKind = (&stTrips == &st[4]) ? 2 : ((&stTrips == &st[2]) ? 1 : 0);
        if (&stTrips == &st[0]) {
            st[6] += nInt;
            st[7] += nPrim;
            st[8] += nIdle;
        }
#endif
        if (nIdle == 64 || (sR > sI && sR > sP)) {
            // ---- retire finished rays, refill idle lanes -------------------------------
            const int ri = isIdle ? (kLean ? riReg : __float_as_int(cold[kColdRi][lane])) : -1;
            if (MODE == 3 && ri >= 0) {
                const int b = ri >> kFusedIndexBits;
                const long idx = ri & ((1 << kFusedIndexBits) - 1);
                void *outp = b == 0 ? p.bOut[0] : (b == 1 ? p.bOut[1] : (b == 2 ? p.bOut[2] : p.bOut[3]));
                const bool needHost = !kLean && p.hasHostPrims && cold[kColdHost][lane] != 0.0f;
                if ((p.anyMask >> b) & 1u) {
                    reinterpret_cast<uint8_t *>(outp)[idx] = found ? 1 : (needHost ? 2 : 0);
                } else {
                    float4 h0, h1;
                    h0.x = cold[kColdHit][lane];
                    h0.y = tMax;
                    h0.z = cold[kColdHit + 1][lane];
                    h0.w = cold[kColdHit + 2][lane];
                    h1.x = cold[kColdHit + 3][lane];
                    h1.y = __int_as_float(visited);
                    h1.z = __int_as_float(tests);
                    h1.w = INST ? cold[kHitInst][lane] : 0.0f;
                    if (needHost) h1.w = __int_as_float(-1);
                    float4 *out = reinterpret_cast<float4 *>(outp) + 2 * idx;
                    out[0] = h0;
                    out[1] = h1;
                }
            } else if (ri >= 0) {
                if (MODE == 0) {
                    float4 h0, h1;
                    h0.x = cold[kColdHit][lane];  // hit primitive id (bit pattern)
                    h0.y = tMax;
                    h0.z = cold[kColdHit + 1][lane];
                    h0.w = cold[kColdHit + 2][lane];
                    h1.x = cold[kColdHit + 3][lane];
                    h1.y = __int_as_float(visited);
                    h1.z = __int_as_float(tests);
                    h1.w = INST ? cold[kHitInst][lane] : 0.0f;  // 0 / instance index + 1 (bit pattern)
                    if (!kLean && p.hasHostPrims && cold[kColdHost][lane] != 0.0f) h1.w = __int_as_float(-1);
                    float4 *out = reinterpret_cast<float4 *>(p.hits) + 2 * (long)ri;
                    out[0] = h0;
                    out[1] = h1;
                } else {
                    const bool needHost = !kLean && p.hasHostPrims && cold[kColdHost][lane] != 0.0f;
                    p.occluded[ri] = found ? 1 : (needHost ? 2 : 0);
                    if (MODE == 1) {
                        if (p.visitedOut) p.visitedOut[ri] = visited;
                        if (p.testsOut) p.testsOut[ri] = tests;
                    }
                }
            }
            int newRi = -1;  // n < 2^31 is enforced by the ABI
            if (exhausted) break;  // only reached with every lane idle (sR == 0 otherwise)
            long start = 0;
            for (;;) {  // find a queue with work left (own XCD's first, then steal)
                if (MODE == 3) {
                    nRays = p.bN[curBatch];
                    if (const int32_t *nd = p.bNDev[curBatch]) {  // wavefront queues: the size lives on the device
                        const long v = *nd;
                        nRays = v < 0 ? 0 : (v < nRays ? v : nRays);
                    }
                }
                // p.nQueues is 1 or kMaxQueues = 8: the division of a long is a shift (the scalar unit has
                // no 64-bit divide; the compiler's expansion was ~250 instructions per refill)
                static_assert(kMaxQueues == 8, "queue ranges are computed with a shift by 3");
                const int qShift = p.nQueues > 1 ? 3 : 0;
                const long qBegin = (nRays * q) >> qShift, qEnd = (nRays * (q + 1)) >> qShift;
                unsigned got = 0;
                if (lane == 0)
                    got = atomicAdd(&p.queue[((MODE == 3 ? curBatch * p.nQueues : 0) + q) * kQueueStrideWords],
                                    (unsigned)nIdle);
                got = __builtin_amdgcn_readfirstlane(got);
                start = qBegin + (long)got;
                if (start < qEnd) {
                    const int rank = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(idleMask >> 32),
                        __builtin_amdgcn_mbcnt_lo((unsigned)idleMask, 0u));
                    if (isIdle && start + rank < qEnd) newRi = (int)(start + rank);
                    break;
                }
                if (++queuesTried >= p.nQueues) {
                    if (MODE == 3 && curBatch + 1 < p.nBatches) {  // this batch is handed out: on to the next
                        ++curBatch;
                        queuesTried = 0;
                        continue;
                    }
                    exhausted = true;
                    break;
                }
                q = (q + 1 == p.nQueues) ? 0 : q + 1;
            }
            if (isIdle) {
                const int tag = (MODE == 3 && newRi >= 0) ? (newRi | (curBatch << kFusedIndexBits)) : newRi;
                if (kLean) riReg = tag;
                else cold[kColdRi][lane] = __int_as_float(tag);
            }
            if (newRi >= 0) {
                const nnbvh_ray *batchRays = MODE == 3 ? p.bRays[curBatch] : p.rays;
                float4 r0, r1;
                if (!SOA || batchRays) {
                    const float4 *in = reinterpret_cast<const float4 *>(batchRays) + 2 * (long)newRi;
                    r0 = in[0];
                    r1 = in[1];
                } else {  // a wavefront queue: SOA<Ray> slices (wavefront/workitems.soa:40-50)
                    const nnbvh_ray_soa &q = MODE == 3 ? p.bSoa[curBatch] : p.soa;
                    r0 = {q.ox[newRi], q.oy[newRi], q.oz[newRi], q.tmax ? q.tmax[newRi] : __builtin_inff()};
                    r1 = {q.dx[newRi], q.dy[newRi], q.dz[newRi], q.time ? q.time[newRi] : 0.0f};
                }
                r.o = {r0.x, r0.y, r0.z};
                tMax = r0.w;
                const V3 d = {r1.x, r1.y, r1.z};
                if (PATCH) {
                    cold[kColdD][lane] = d.x;
                    cold[kColdD + 1][lane] = d.y;
                    cold[kColdD + 2][lane] = d.z;
                }
                // aggregates.cpp:534-535
                r.inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
                ray_shear(r, d);
                if (MODE == 0 || MODE == 3) {
                    cold[kColdHit][lane] = __int_as_float(-1);
                    cold[kColdHit + 1][lane] = 0.0f;
                    cold[kColdHit + 2][lane] = 0.0f;
                    cold[kColdHit + 3][lane] = 0.0f;
                }
                if (INST) {
                    cold[kHitInst][lane] = 0.0f;
                    cold[kCurInst][lane] = 0.0f;
                    cold[kColdTime][lane] = r1.w;  // ray.time: read by animated instances
                    floor = -1;
                }
                if (!kLean && p.hasHostPrims) cold[kColdHost][lane] = 0.0f;
                visited = 1;  // the root
                tests = 0;
                found = false;
                sp = base = 0;
                float tEntry;
                const bool rootHit =
                    slab_partial(p.rootMin[0], p.rootMin[1], p.rootMin[2], p.rootMax[0],
                                 p.rootMax[1], p.rootMax[2], r, tEntry) &&
                    (tEntry < tMax);
                cur = rootHit ? p.rootRef : kDone;
            }
            if (kLean && NNBVH_MASKS) masks = ray_masks(r);  // some lanes carry new rays
            continue;
        }

        if (INST == 2 && NNBVH_ANIM_ENTER_WEIGHT > 0 && nEnter > 0 &&
            ((nInt == 0 && nPrim == 0) || (nEnter * NNBVH_ANIM_ENTER_WEIGHT > sI && nEnter * NNBVH_ANIM_ENTER_WEIGHT > sP))) {
            // ---- enter step: every lane waiting at an AnimatedPrimitive enters it now -------------------
            if (cur == kEnter) {
                const int slot = __float_as_int(cold[kPendSlot][lane]);
                const float4 s0 = p.prims[slot], s1 = p.prims[slot + 1], s2 = p.prims[slot + 2];
                enter_instance(slot, __float_as_uint(s1.w), s0, s1, s2);
            }
            continue;
        }

        if (kLean && NNBVH_MERGED && p.primMin > 0) {
            // ---- merged trip (lean instances): every lane with a node pending, and — when at least
            // p.primMin lanes wait on a leaf, or nobody has a node — every lane with a primitive pending,
            // fetches 64 B through ONE load sequence (records and slots live in one allocation: a lane's
            // 32-bit offset reaches either), then the two kinds of arithmetic follow each other.  A trip
            // costs one round trip to memory whichever lanes take part; lanes waiting on a leaf no longer
            // sit out their wavefront's interior trips.  (A primitive lane's fourth slot is not used.)
            const bool primLane = !isInt && !isIdle && (nPrim >= p.primMin || nInt == 0);
#ifdef NNBVH_STATS
            st[0] += 1;
            st[1] += nInt;
            if (nPrim >= p.primMin || nInt == 0) {
                st[2] += 1;
                st[3] += nPrim;
            }
            stKind = 0;
#endif
            if (isInt || primLane) {
                const unsigned off = isInt ? ((unsigned)cur << 6) : (p.primsOff + ((unsigned)~cur << 4));
                const float4 *rec = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.wide) + off);
                const float4 q3 = rec[3];
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                if (isInt) interior_math(q0, q1, q2, q3);
                else prim_math(~cur, q0, q1, q2);
            }
            for (int rep = 1; rep < p.intRepeat && __ballot(cur >= 0) != 0ull; ++rep) interior_step();
            continue;
        }

        if (sP > sI || nInt == 0) {
            // ---- primitive step: lanes with a pending leaf test ONE primitive -----------
            // (up to p.primRepeat of them per scheduling decision: lanes whose leaf is finished sit the
            // rest out, lanes still inside theirs go on without another round of ballots)
            // (the one-launch kernel and the lean instances keep ONE: compiled in, the loop costs them 6
            // registers — spills at 8 wavefronts per SIMD — and 2.6 % / 8 % of their rate)
            const int nPrep = ((MODE == 3 && !NNBVH_FUSED_PRIM_LOOP) || (kLean && !NNBVH_LEAN_PRIM_LOOP)) ? 1 : p.primRepeat;
            int prep = 0;
            do {
            if (cur < 0 && cur != kDone) {
                if (INST && cur == kReturn) leave_instance();
                if (cur < 0 && cur != kDone && cur != kReturn && (INST != 2 || cur != kEnter)) {
                    const int slot = ~cur;
                    const float4 *pr = kLean ? reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.prims) + ((unsigned)slot << 4))
                                             : p.prims + slot;
                    float4 s0 = pr[0], s1 = pr[1], s2 = pr[2];
                    // all three slots are fetched before anything looks at the flags: left alone the compiler
                    // loads the flag word first and the vertices only inside the not-degenerate branch, two
                    // dependent trips to memory per primitive
                    asm volatile("" : "+v"(s0.x), "+v"(s0.y), "+v"(s0.z), "+v"(s0.w), "+v"(s1.x), "+v"(s1.y),
                                      "+v"(s1.z), "+v"(s1.w), "+v"(s2.x), "+v"(s2.y), "+v"(s2.z), "+v"(s2.w));
                    prim_math(slot, s0, s1, s2);
                }
            }
            } while (++prep < nPrep && __ballot(cur < 0 && cur != kDone) != 0ull);
        } else {
            // ---- interior step(s): up to p.intRepeat in a row before the next scheduling
            // decision (lanes that leave the interior state sit the remaining ones out) -------
            if (MODE == 3) {
                // the one-launch kernel: the first three steps (the default int_repeat) written out, no trip
                // counter to maintain (+0.8 %; the separate-launch kernels lose 1.8 % with it and keep the loop)
                interior_step();
                if (p.intRepeat > 1 && __ballot(cur >= 0) != 0ull) {
                    interior_step();
                    if (p.intRepeat > 2 && __ballot(cur >= 0) != 0ull) {
                        interior_step();
                        for (int rep = 3; rep < p.intRepeat && __ballot(cur >= 0) != 0ull; ++rep) interior_step();
                    }
                }
            } else {
            int rep = 0;
            do {
                interior_step();
            } while (++rep < p.intRepeat && __ballot(cur >= 0) != 0ull);
            }
        }
    }
#ifdef NNBVH_STATS
    if (lane == 0 && p.stats)
        for (int k = 0; k < 16; ++k) atomicAdd(&p.stats[k], st[k]);
#endif
}

// ------------------------------------------------------------------------------------
template <int MODE, int W, int INST, int PATCH, int ALPHA = 0, int SOA = 0>
static hipError_t launch_one(const TraceParams &p, int blocks, hipStream_t stream, int *occupancy) {
    if (occupancy) {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, trace_kernel<MODE, W, INST, PATCH, ALPHA, SOA>,
                                                            kBlockThreads, 0);
    }
    hipLaunchKernelGGL((trace_kernel<MODE, W, INST, PATCH, ALPHA, SOA>), dim3((unsigned)blocks), dim3(kBlockThreads), 0,
                       stream, p);
    return hipGetLastError();
}

// mode 3 exists for the window-8 instances without alpha-tested triangles
static hipError_t launch_fused(const TraceParams &p, int window, int instanced, int patches, int blocks,
                               hipStream_t stream, int *occupancy) {
    if (window != 8 || (patches & 2)) return hipErrorInvalidValue;
    if (instanced) return p.anim ? launch_one<3, 8, 2, 1>(p, blocks, stream, occupancy)
                                 : launch_one<3, 8, 1, 1>(p, blocks, stream, occupancy);
    bool soa = false;
    for (int b = 0; b < p.nBatches; ++b) soa = soa || !p.bRays[b];
    if (!patches && !p.hasHostPrims && p.fits32)
        return soa ? launch_one<3, 8, 0, 0, 0, 1>(p, blocks, stream, occupancy) : launch_one<3, 8, 0, 0>(p, blocks, stream, occupancy);
    if (soa) return hipErrorInvalidValue;  // SOA batches: lean instances only (the caller gathers otherwise)
    return launch_one<3, 8, 0, 1>(p, blocks, stream, occupancy);
}

template <int MODE>
static hipError_t launch_mode(const TraceParams &p, int window, int instanced, int patches, int blocks,
                              hipStream_t stream, int *occupancy) {
    // scenes with alpha-tested triangles and two-level scenes: one instance of the kernel each (window 8)
    // INST: 0 single-level, 1 static instances, 2 instances with AnimatedPrimitives among them (the
    // interpolation of the transform costs 60 VGPRs: 160-177 against 98-116, 2 against 3 wavefronts per SIMD)
    if (patches & 4) {  // alpha-tested bilinear patches
        if (!instanced) return launch_one<MODE, 8, 0, 1, 2>(p, blocks, stream, occupancy);
        return p.anim ? launch_one<MODE, 8, 2, 1, 2>(p, blocks, stream, occupancy)
                      : launch_one<MODE, 8, 1, 1, 2>(p, blocks, stream, occupancy);
    }
    if (patches & 2) {
        if (!instanced) return launch_one<MODE, 8, 0, 1, 1>(p, blocks, stream, occupancy);
        return p.anim ? launch_one<MODE, 8, 2, 1, 1>(p, blocks, stream, occupancy)
                      : launch_one<MODE, 8, 1, 1, 1>(p, blocks, stream, occupancy);
    }
    if (instanced) return p.anim ? launch_one<MODE, 8, 2, 1>(p, blocks, stream, occupancy)
                                 : launch_one<MODE, 8, 1, 1>(p, blocks, stream, occupancy);
    const bool soa = !p.rays && !occupancy;
    if (!patches && !p.hasHostPrims && p.fits32 && window == 8 && !instanced) {
        if (soa && MODE != 1) return launch_one<MODE == 1 ? 0 : MODE, 8, 0, 0, 0, 1>(p, blocks, stream, occupancy);
        if (!soa) return launch_one<MODE, 8, 0, 0>(p, blocks, stream, occupancy);
    }
    if (soa) return hipErrorInvalidValue;  // SOA batches: lean instances of modes 0 / 2 / 3 only
    switch (window) {
    case 4: return launch_one<MODE, 4, 0, 1>(p, blocks, stream, occupancy);
    case 8: return launch_one<MODE, 8, 0, 1>(p, blocks, stream, occupancy);
    case 16: return launch_one<MODE, 16, 0, 1>(p, blocks, stream, occupancy);
    default: return hipErrorInvalidValue;
    }
}

// The queue heads are cleared by a (tiny) kernel rather than hipMemsetAsync: as a kernel node the
// clear keeps its place between the launches of a captured hipGraph.
__global__ void zero_queue_kernel(unsigned *queue, int words) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) queue[i] = 0u;
}
hipError_t launch_zero_queue(unsigned *queue, int words, hipStream_t stream) {
    hipLaunchKernelGGL(zero_queue_kernel, dim3(1), dim3(256), 0, stream, queue, words);
    return hipGetLastError();
}

// occupancy != nullptr: no launch, only report resident blocks per CU for that instance
hipError_t launch_trace(int mode, const TraceParams &p, int window, int instanced, int patches, int blocks,
                        hipStream_t stream, int *occupancy) {
    switch (mode) {
    case 0: return launch_mode<0>(p, window, instanced, patches, blocks, stream, occupancy);
    case 1: return launch_mode<1>(p, window, instanced, patches, blocks, stream, occupancy);
    case 2: return launch_mode<2>(p, window, instanced, patches, blocks, stream, occupancy);
    case 3: return launch_fused(p, window, instanced, patches, blocks, stream, occupancy);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace nnbvh
