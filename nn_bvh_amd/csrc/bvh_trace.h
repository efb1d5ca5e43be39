// bvh_trace.h — launch interface between the C ABI (bvh_capi.cpp) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnbvh.h"
#include "nnbvh_internal.h"

namespace nnbvh {

constexpr int kBlockThreads = 256;   // 4 wavefronts
constexpr int kMaxQueues = 8;        // one ray queue per XCD
constexpr int kQueueStrideWords = 32;  // each queue head on its own 128-B line
constexpr int kMaxFusedBatches = 4;    // batches one mode-3 launch may cover
constexpr int kFusedIndexBits = 28;    // a lane's ray tag = batch << 28 | index: batches below 2^28 rays

struct TraceParams {
    const float4 *wide;   // interior records, 4 x float4 each
    const float4 *prims;  // prim stream, 16-B slots
    float rootMin[3], rootMax[3];
    int rootRef;
    const nnbvh_ray *rays;
    nnbvh_hit *hits;        // MODE 0
    uint8_t *occluded;      // MODE 1/2
    int32_t *visitedOut;    // MODE 1, nullable
    int32_t *testsOut;      // MODE 1, nullable
    long n;
    const int32_t *nDev;    // nullable: device-resident batch size, clamped to [0, n]
    unsigned *queue;        // nQueues heads, kQueueStrideWords apart, zeroed before launch
    int nQueues;
    int primWeight;         // scheduling weight of a lane waiting on a primitive (interior = 16)
    int refillWeight;       // scheduling weight of an idle lane (interior = 16)
    int hasHostPrims;       // scene contains NNBVH_PRIM_HOST primitives
    int intRepeat;          // interior steps per scheduling decision (>= 1)
    int primRepeat;         // primitive steps per scheduling decision (>= 1)
    int fits32;             // wide[] and prims[] are both below 4 GiB: the lean instances' 32-bit offsets reach them
    unsigned primsOff;      // byte offset of prims[] from wide[] (one allocation; lean instances, merged trips)
    int primMin;            // merged trips: lanes that must wait on a leaf before the primitive block runs; 0 = separate trips
    unsigned long long *stats;  // NNBVH_STATS builds: trips/lanes per step kind; else unused
    uint2 *spill;           // [kMaxStack][grid threads] overflow of the LDS stack window
    const float *anim;      // two-level scenes: kAnimStride floats per instance (anim_math.h), or null
    // mode 3 (one launch over several batches, closest-hit and occlusion-only any-hit mixed): batch
    // b's rays / results / size, queue heads at queue[(b * nQueues + q) * kQueueStrideWords]; bit b
    // of anyMask = batch b is any-hit (bOut = uint8 occluded[]), else closest (bOut = nnbvh_hit[])
    int nBatches;
    unsigned anyMask;
    const nnbvh_ray *bRays[kMaxFusedBatches];
    void *bOut[kMaxFusedBatches];
    long bN[kMaxFusedBatches];
    const int32_t *bNDev[kMaxFusedBatches];  // nullable: device-resident size of batch b, clamped to [0, bN[b]]
    // rays == nullptr (bRays[b] == nullptr): the batch is a wavefront queue, read as the SOA<Ray> slices it is
    // (tmax == nullptr: Infinity, time == nullptr: 0) — no gather pass into nnbvh_ray records
    nnbvh_ray_soa soa;
    nnbvh_ray_soa bSoa[kMaxFusedBatches];
};

// bvh_layout.cpp: re-orders the baked arrays in memory (speed only; see the modes there)
bool relayout_scene(int mode, float4 **d_wide, float4 **d_prims, int *n_interior, int64_t *n_slots, int *root_ref,
                    int top_levels, std::string *error);

hipError_t launch_zero_queue(unsigned *queue, int words, hipStream_t stream);

// occupancy != nullptr: do not launch, report resident blocks per CU of that kernel instance
// patches: bit 0 = the scene holds bilinear patches (0: kernel instances without the parked ray direction),
// bit 1 = it holds alpha-tested triangles (the ALPHA kernel instances)
hipError_t launch_trace(int mode, const TraceParams &p, int window, int instanced, int patches, int blocks,
                        hipStream_t stream, int *occupancy);

}  // namespace nnbvh
