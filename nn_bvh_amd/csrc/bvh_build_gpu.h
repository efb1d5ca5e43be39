// bvh_build_gpu.h — interface between the host builder (bvh_build.cpp) and the device HLBVH
// pipeline (bvh_build_gpu.hip).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/nnbvh.h"

namespace nnbvh {

// Upper levels of an HLBVH (buildUpperSAH, cpu/aggregates.cpp:626-723) over treelet roots given
// by bounds (6 floats each) and subtree node counts, laid out in flattenBVH's DFS order
// (aggregates.cpp:505-522): where every treelet's nodes start, how deep its root sits, and the
// upper interior nodes themselves.
struct UpperLayout {
    std::vector<int> base;    // [nTreelets] flat index of the treelet's root
    std::vector<int> depth;   // [nTreelets] depth of the treelet's root
    std::vector<int> upper_index;                 // flat indices of the upper interior nodes
    std::vector<nnbvh_linear_node> upper_nodes;   // ... and their contents
    int total_nodes = 0;
};
bool hlbvh_upper_layout(const float *treelet_bounds, const int *treelet_sizes, int n_treelets,
                        UpperLayout *out, std::string *error);

struct GpuBuildResult {
    std::vector<nnbvh_linear_node> nodes;
    std::vector<nnbvh_prim> ordered;
    int depth = 0;
    // milliseconds: host->device copies, device pipeline up to the treelet table, host upper tree,
    // device emit, device->host copies
    double ms[5] = {0, 0, 0, 0, 0};
    int n_treelets = 0, n_unique_codes = 0;
    // keep_on_device (in): do not download; hand the device arrays to the caller instead (who frees
    // them with hipFree): LinearBVHNode[total_nodes], leaf-ordered nnbvh_prim[n_prims], float3 verts
    bool keep_on_device = false;
    void *d_nodes = nullptr, *d_ordered = nullptr, *d_verts = nullptr;
    int total_nodes = 0;
};
// prim_bounds may be NULL when every primitive is a triangle or a bilinear patch.
bool gpu_hlbvh(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
               const float *prim_bounds, int max_prims_in_node, int device, GpuBuildResult *out,
               std::string *error);

// SAH (buildRecursive's default branch) on the device; same tree and leaf order as the host builder.
// ms: upload, big nodes breadth-first, subtrees (one wavefront each), layout + bounds, download;
// n_treelets = subtrees built by wavefronts, n_unique_codes = nodes built breadth-first.
bool gpu_sah(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts, const float *prim_bounds,
             int max_prims_in_node, int device, GpuBuildResult *out, std::string *error);

// Device-side counterpart of nnbvh_scene_create's baking (bvh_capi.cpp): the 64-B "both children"
// records and the 16-B-slot primitive stream, straight from device-resident build output.  Triangles,
// bilinear patches and host-only primitives (no instances).
struct BakedScene {
    void *d_wide = nullptr, *d_prims = nullptr;  // float4 arrays, owned by the caller afterwards
    int n_interior = 0;
    long n_slots = 0;
    int root_ref = 0;
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    int has_host_prims = 0;
    int has_patches = 0;
    int has_alpha = 0;
};
// d_normals: per-vertex shading normals (3 floats, indexed like the vertices) or null; read for
// NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH primitives only
bool bake_on_device(const void *d_nodes, int n_nodes, const void *d_ordered_prims, int n_prims, const void *d_verts,
                    int device, BakedScene *out, std::string *error, const void *d_normals = nullptr,
                    const void *d_prim_alpha = nullptr, const void *d_uvs = nullptr);

// d_ordered: leaf-ordered nnbvh_prim[n_prims] of a build that ran with ids = positions in the caller's array;
// gathers prim_alpha into that order (*d_alpha_out, hipMalloc'ed, the caller's to free) and restores caller_ids
bool gather_prim_alpha_on_device(void *d_ordered, int n_prims, const float *prim_alpha, const int32_t *caller_ids,
                                 void **d_alpha_out, std::string *error);

}  // namespace nnbvh
