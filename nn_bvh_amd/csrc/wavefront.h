// wavefront.h — launch interface of the wavefront-queue kernels (wavefront.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnbvh.h"

namespace nnbvh {

struct WavefrontCount {
    int n;               // host bound (maxRays)
    const int32_t *nDev; // nullable device-resident queue size, clamped to [0, n]
};

hipError_t launch_wf_gather(const nnbvh_ray_soa &q, WavefrontCount cnt, void *rays, int maxBlocks,
                            hipStream_t stream);
hipError_t launch_wf_enqueue_closest(const void *hits, WavefrontCount cnt, const uint8_t *hasMedium,
                                     const uint8_t *primClass, long nPrimClass,
                                     const nnbvh_closest_queues &out, int maxBlocks,
                                     hipStream_t stream);
hipError_t launch_wf_record_shadow(const uint8_t *occluded, WavefrontCount cnt, const float *Ld,
                                   const float *ru, const float *rl, const int32_t *pixelIndex,
                                   float *L, long nPixels, int maxBlocks, hipStream_t stream);

}  // namespace nnbvh
