// kd_build_gpu.h — interface between the kd-tree C ABI (kd_build.cpp) and the device builder (kd_build_gpu.hip).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/nnbvh.h"

namespace nnbvh {

struct KdGpuResult {
    std::vector<nnbvh_kd_node> nodes;
    std::vector<int32_t> prim_indices;
    int depth = 0;
    int levels = 0;
    double device_ms = 0;  // upload of the primitive bounds .. the two output arrays written on the device
    double total_ms = 0;   // ... and downloaded
};

// prim_bounds: 6 floats (min, max) per primitive, finite (the caller validated them); bounds = their union.
// Same node array as the host builder; primitives inside multi-primitive leaves in std::stable_sort order
// (see the note at the top of kd_build_gpu.hip).
bool gpu_kd_build(const float *prim_bounds, int n_prims, const float bounds[6], int isect_cost, int traversal_cost,
                  float empty_bonus, int max_prims, int max_depth, int device, KdGpuResult *out, std::string *error);

}  // namespace nnbvh
