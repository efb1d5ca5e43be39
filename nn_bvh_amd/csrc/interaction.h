// interaction.h — launch interface of the hit -> SurfaceInteraction post-pass (interaction.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnbvh.h"

namespace nnbvh {

struct ShadingMeshDevice {  // device pointers of a nnbvh_shading_mesh
    float *verts = nullptr;
    int32_t *triVerts = nullptr, *patchVerts = nullptr;
    float *normals = nullptr, *uvs = nullptr, *tangents = nullptr;
    int32_t *faceIndices = nullptr;
    uint8_t *triFlags = nullptr;
    int nTris = 0, nVerts = 0;
    nnbvh_instance *instances = nullptr;  // optional: hits inside instances are finished on the device too
    int nInstances = 0;
    // optional: AnimatedPrimitive instances — the traversal kernels' table (kAnimStride floats per instance,
    // anim_math.h) and rows 0..2 of startTransform.m / endTransform.m (24 floats per instance)
    float *anim = nullptr, *animFwd = nullptr;
    unsigned defaultFlags = 0;
};

hipError_t launch_triangle_interactions(const ShadingMeshDevice &m, const void *rays, const nnbvh_ray_soa *soa,
                                        const void *hits, int n, const int32_t *nDev, void *out, int maxBlocks,
                                        hipStream_t stream);

}  // namespace nnbvh
