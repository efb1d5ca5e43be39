// wavefront2.h — launch interface of the kernels behind IntersectShadowTr / IntersectOneRandom
// (wavefront2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnbvh.h"
#include "wavefront.h"

namespace nnbvh {

// ---- IntersectShadowTr ------------------------------------------------------------------------
hipError_t launch_str_init(const nnbvh_ray_soa &q, WavefrontCount cnt, void *rays, int32_t *orig, float4 *pLight,
                           uint8_t *state, int maxBlocks, hipStream_t stream);
// hits of the current rays -> per-item verdicts; rays that hit an interface surface are appended
// (ray, hit, item) to the next list, *counter counting them
hipError_t launch_str_classify(const void *raysCur, const void *hitsCur, const int32_t *origCur,
                               const int32_t *nCur, const uint8_t *primClass, long nPrimClass, uint8_t *state,
                               void *raysNext, void *hitsNext, int32_t *origNext, int32_t *counter, int maxItems,
                               int maxBlocks, hipStream_t stream);
// interactions of the listed hits -> the next segment's ray towards pLight (compacted into raysOut)
hipError_t launch_str_spawn(const void *raysIn, const void *intr, const int32_t *origIn, const int32_t *nIn,
                            const float4 *pLight, uint8_t *state, void *raysOut, int32_t *origOut,
                            int32_t *counter, int maxItems, int maxBlocks, hipStream_t stream);
hipError_t launch_str_record(const uint8_t *state, WavefrontCount cnt, const float *Ld, const float *ru,
                             const float *rl, const int32_t *pixelIndex, float *L, long nPixels,
                             uint8_t *visibleOut, int maxBlocks, hipStream_t stream);

// ---- IntersectOneRandom -----------------------------------------------------------------------
struct OneRandomState {  // per work item, device arrays
    float *pi;           // 9 floats per item: pi low[3], high[3], n[3] of the current base interaction
    uint64_t *rng;       // 2 per item: PCG32 state, inc
    float *weights;      // 2 per item: weightSum, reservoirWeight
};
hipError_t launch_or_init(const float *p0, const float *p1, WavefrontCount cnt, OneRandomState st, void *raysOut,
                          int32_t *origOut, int32_t *counter, void *selHits, void *selRays, int maxBlocks,
                          hipStream_t stream);
hipError_t launch_or_step(const void *raysCur, const void *hitsCur, const void *intrCur, const int32_t *origCur,
                          const int32_t *nCur, const float *p1, const int32_t *material,
                          const int32_t *primMaterial, long nPrimMaterial, OneRandomState st, void *raysNext,
                          int32_t *origNext, int32_t *counter, void *selHits, void *selRays, int maxItems,
                          int maxBlocks, hipStream_t stream);
hipError_t launch_or_finish(WavefrontCount cnt, OneRandomState st, float *pdf, float *weightSum, int maxBlocks,
                            hipStream_t stream);

}  // namespace nnbvh
