// bvh_bake.hip — the traversal kernels' data layout (nnbvh_internal.h) produced on the device from
// a device-resident LinearBVHNode array + leaf-ordered primitive table: the counterpart of the
// host baking in bvh_capi.cpp (create_scene), bit for bit, for scenes of triangles, bilinear
// patches and host-only primitives.  With the device builders (bvh_build_gpu.hip) a scene goes
// from triangles to traceable without its tree ever visiting the host.
//
// Four streaming passes over 24-32-B records plus three prefix sums; HBM-bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>

#include <rocprim/device/device_scan.hpp>

#include "bvh_build_gpu.h"
#include "nnbvh_internal.h"

namespace nnbvh {
namespace {

constexpr int kBk = 256;

__global__ __launch_bounds__(kBk) void k_bake_slot_counts(const nnbvh_prim *__restrict__ prims, int n,
                                                         int *__restrict__ slots, int *flags) {
    const int i = blockIdx.x * kBk + threadIdx.x;
    if (i >= n) return;
    const int kind = prims[i].kind;
    int c = 3;  // triangle, host-only primitive
    if (kind == NNBVH_PRIM_BILINEAR_PATCH) {
        c = 4;
        atomicOr(flags, 4);
    } else if (kind == NNBVH_PRIM_HOST) atomicOr(flags, 1);
    else if (kind == NNBVH_PRIM_ALPHA_TRIANGLE || kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED) atomicOr(flags, 8);
    else if (kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH || kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED) {
        c = 6;  // three slots of vertex normals follow the triangle's
        atomicOr(flags, 8 | 16);
    }
    else if (is_alpha_patch_kind(kind)) {
        c = alpha_patch_slots(kind);  // four slots of vertex normals and / or two of (u, v) follow the patch's
        atomicOr(flags, 4 | 8 | 32 | (is_smooth_alpha_patch_kind(kind) ? 16 : 0) | (is_uv_alpha_patch_kind(kind) ? 64 : 0));
    }
    else if (kind != NNBVH_PRIM_TRIANGLE) atomicOr(flags, 2);  // instances are not baked here
    slots[i] = c;
}

__global__ __launch_bounds__(kBk) void k_bake_node_flags(const nnbvh_linear_node *__restrict__ nodes, int n,
                                                        int *__restrict__ interior,
                                                        unsigned char *__restrict__ leafLast) {
    const int i = blockIdx.x * kBk + threadIdx.x;
    if (i >= n) return;
    const nnbvh_linear_node nd = nodes[i];
    interior[i] = nd.nprims == 0 ? 1 : 0;
    if (nd.nprims != 0) leafLast[nd.offset + nd.nprims - 1] = 1;
}

// DifferenceOfProducts (util/math.h:569-575) and the degenerate-triangle test of shapes.cpp:176-177
__device__ __forceinline__ float bake_dop(float a, float b, float c, float d) {
    const float cd = c * d;
    const float diff = __builtin_fmaf(a, b, -cd);
    const float err = __builtin_fmaf(-c, d, cd);
    return diff + err;
}

__global__ __launch_bounds__(kBk) void k_bake_stream(const nnbvh_prim *__restrict__ prims, int n,
                                                    const float *__restrict__ verts,
                                                    const float *__restrict__ normals,
                                                    const float *__restrict__ uvs,
                                                    const float *__restrict__ primAlpha,
                                                    const int *__restrict__ slotOf,
                                                    const unsigned char *__restrict__ leafLast,
                                                    float4 *__restrict__ stream) {
    const int i = blockIdx.x * kBk + threadIdx.x;
    if (i >= n) return;
    const nnbvh_prim p = prims[i];
    float4 *s = stream + slotOf[i];
    unsigned flags = leafLast[i] ? kPrimLast : 0u;
    if (p.kind == NNBVH_PRIM_HOST) {
        flags |= kPrimHost;
        s[0] = make_float4(0, 0, 0, __int_as_float(p.id));
        s[1] = make_float4(0, 0, 0, __uint_as_float(flags));
        s[2] = make_float4(0, 0, 0, 0);
        return;
    }
    const int nv = (p.kind == NNBVH_PRIM_BILINEAR_PATCH || is_alpha_patch_kind(p.kind)) ? 4 : 3;
    float v[4][3];
    for (int j = 0; j < nv; ++j)
        for (int k = 0; k < 3; ++k) v[j][k] = verts[3 * (long)p.v[j] + k];
    if (nv == 4) {
        flags |= kPrimPatch;
    } else {
        const float ax = v[2][0] - v[0][0], ay = v[2][1] - v[0][1], az = v[2][2] - v[0][2];
        const float bx = v[1][0] - v[0][0], by = v[1][1] - v[0][1], bz = v[1][2] - v[0][2];
        const float cx = bake_dop(ay, bz, az, by), cy = bake_dop(az, bx, ax, bz), cz = bake_dop(ax, by, ay, bx);
        if (cx * cx + cy * cy + cz * cz == 0.0f) flags |= kPrimDegenerate;
    }
    float alpha = 0.0f;
    if (p.kind == NNBVH_PRIM_ALPHA_TRIANGLE || p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED) {
        flags |= kPrimAlpha | (p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED ? kPrimFlipN : 0u);
        alpha = __int_as_float(p.v[3]);
    }
    if (p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH || p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED) {
        flags |= kPrimAlpha | kPrimSmooth | (p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED ? kPrimFlipN : 0u);
        alpha = __int_as_float(p.v[3]);
        for (int j = 0; j < 3; ++j)
            s[3 + j] = make_float4(normals[3 * (long)p.v[j]], normals[3 * (long)p.v[j] + 1], normals[3 * (long)p.v[j] + 2], 0);
    }
    if (is_alpha_patch_kind(p.kind)) {
        flags |= kPrimAlpha | (is_flipped_alpha_patch_kind(p.kind) ? kPrimFlipN : 0u);
        alpha = primAlpha[i];
        if (is_smooth_alpha_patch_kind(p.kind)) {
            flags |= kPrimSmooth;
            for (int j = 0; j < 4; ++j)
                s[4 + j] = make_float4(normals[3 * (long)p.v[j]], normals[3 * (long)p.v[j] + 1], normals[3 * (long)p.v[j] + 2], 0);
        }
        if (is_uv_alpha_patch_kind(p.kind)) {
            flags |= kPrimUV;
            float4 *u = s + (is_smooth_alpha_patch_kind(p.kind) ? 8 : 4);
            u[0] = make_float4(uvs[2 * (long)p.v[0]], uvs[2 * (long)p.v[0] + 1], uvs[2 * (long)p.v[1]], uvs[2 * (long)p.v[1] + 1]);
            u[1] = make_float4(uvs[2 * (long)p.v[2]], uvs[2 * (long)p.v[2] + 1], uvs[2 * (long)p.v[3]], uvs[2 * (long)p.v[3] + 1]);
        }
    }
    s[0] = make_float4(v[0][0], v[0][1], v[0][2], __int_as_float(p.id));
    s[1] = make_float4(v[1][0], v[1][1], v[1][2], __uint_as_float(flags));
    s[2] = make_float4(v[2][0], v[2][1], v[2][2], alpha);
    if (nv == 4) s[3] = make_float4(v[3][0], v[3][1], v[3][2], 0);
}

__device__ __forceinline__ int bake_ref(const nnbvh_linear_node &nd, int index, const int *ord, const int *slotOf) {
    return nd.nprims == 0 ? ord[index] : ~slotOf[nd.offset];
}

__global__ __launch_bounds__(kBk) void k_bake_wide(const nnbvh_linear_node *__restrict__ nodes, int n,
                                                  const int *__restrict__ ord, const int *__restrict__ slotOf,
                                                  float4 *__restrict__ wide) {
    const int i = blockIdx.x * kBk + threadIdx.x;
    if (i >= n) return;
    const nnbvh_linear_node nd = nodes[i];
    if (nd.nprims != 0) return;
    const nnbvh_linear_node c0 = nodes[i + 1], c1 = nodes[nd.offset];
    float4 *w = wide + 4 * (long)ord[i];
    w[0] = make_float4(c0.pmin[0], c0.pmin[1], c0.pmin[2], c0.pmax[0]);
    w[1] = make_float4(c0.pmax[1], c0.pmax[2], c1.pmin[0], c1.pmin[1]);
    w[2] = make_float4(c1.pmin[2], c1.pmax[0], c1.pmax[1], c1.pmax[2]);
    w[3] = make_float4(__int_as_float(bake_ref(c0, i + 1, ord, slotOf)), __int_as_float(bake_ref(c1, nd.offset, ord, slotOf)),
                       __int_as_float((int)nd.axis), __int_as_float(0));
}

struct Scratch {
    void *ptrs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    void *get(size_t bytes) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        ptrs[n++] = p;
        return p;
    }
    ~Scratch() {
        for (int i = 0; i < n; ++i) (void)hipFree(ptrs[i]);
    }
};

// Device-built scenes with a per-primitive attribute: the build ran with ids = positions in the caller's array, so
// the ordered primitives say where each one came from; gather the attribute and put the caller's ids back.
__global__ __launch_bounds__(kBk) void k_gather_alpha_restore_ids(nnbvh_prim *__restrict__ ordered, int n,
                                                                 const float *__restrict__ alphaIn,
                                                                 const int *__restrict__ callerIds,
                                                                 float *__restrict__ alphaOut) {
    const int i = blockIdx.x * kBk + threadIdx.x;
    if (i >= n) return;
    const int from = ordered[i].id;
    alphaOut[i] = alphaIn[from];
    ordered[i].id = callerIds[from];
}

}  // namespace

#define BK_CHECK(expr, what)                                                            \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            *error = std::string("device bake: ") + what + ": " + hipGetErrorString(e_); \
            return false;                                                               \
        }                                                                               \
    } while (0)

bool bake_on_device(const void *d_nodes_, int n_nodes, const void *d_prims_, int n_prims, const void *d_verts_,
                    int device, BakedScene *out, std::string *error, const void *d_normals_, const void *d_prim_alpha_,
                    const void *d_uvs_) {
    const auto *dNodes = (const nnbvh_linear_node *)d_nodes_;
    const auto *dPrims = (const nnbvh_prim *)d_prims_;
    const auto *dVerts = (const float *)d_verts_;
    BK_CHECK(hipSetDevice(device), "hipSetDevice");
    hipStream_t stream = nullptr;
    Scratch sc;
    int *dSlots = (int *)sc.get(((size_t)n_prims + 1) * sizeof(int));
    int *dSlotOf = (int *)sc.get(((size_t)n_prims + 1) * sizeof(int));
    int *dInterior = (int *)sc.get(((size_t)n_nodes + 1) * sizeof(int));
    int *dOrd = (int *)sc.get(((size_t)n_nodes + 1) * sizeof(int));
    unsigned char *dLeafLast = (unsigned char *)sc.get((size_t)n_prims);
    int *dFlags = (int *)sc.get(sizeof(int));
    size_t scanBytes = 0;
    const size_t scanN = (size_t)std::max(n_prims, n_nodes) + 1;
    BK_CHECK(rocprim::exclusive_scan(nullptr, scanBytes, dSlots, dSlotOf, 0, scanN, rocprim::plus<int>(), stream),
             "scan (size query)");
    void *dTmp = sc.get(scanBytes);
    if (!dSlots || !dSlotOf || !dInterior || !dOrd || !dLeafLast || !dFlags || !dTmp) {
        *error = "device bake: hipMalloc failed";
        return false;
    }
    BK_CHECK(hipMemsetAsync(dSlots, 0, ((size_t)n_prims + 1) * sizeof(int), stream), "memset");
    BK_CHECK(hipMemsetAsync(dInterior, 0, ((size_t)n_nodes + 1) * sizeof(int), stream), "memset");
    BK_CHECK(hipMemsetAsync(dLeafLast, 0, (size_t)n_prims, stream), "memset");
    BK_CHECK(hipMemsetAsync(dFlags, 0, sizeof(int), stream), "memset");
    const int gp = (n_prims + kBk - 1) / kBk, gn = (n_nodes + kBk - 1) / kBk;
    hipLaunchKernelGGL(k_bake_slot_counts, dim3(gp), dim3(kBk), 0, stream, dPrims, n_prims, dSlots, dFlags);
    hipLaunchKernelGGL(k_bake_node_flags, dim3(gn), dim3(kBk), 0, stream, dNodes, n_nodes, dInterior, dLeafLast);
    BK_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dSlots, dSlotOf, 0, (size_t)n_prims + 1, rocprim::plus<int>(), stream),
             "scan slots");
    BK_CHECK(rocprim::exclusive_scan(dTmp, scanBytes, dInterior, dOrd, 0, (size_t)n_nodes + 1, rocprim::plus<int>(), stream),
             "scan interior nodes");
    int nSlots = 0, nInterior = 0, flags = 0;
    nnbvh_linear_node root;
    BK_CHECK(hipMemcpyAsync(&nSlots, dSlotOf + n_prims, sizeof(int), hipMemcpyDeviceToHost, stream), "read slot count");
    BK_CHECK(hipMemcpyAsync(&nInterior, dOrd + n_nodes, sizeof(int), hipMemcpyDeviceToHost, stream), "read interior count");
    BK_CHECK(hipMemcpyAsync(&flags, dFlags, sizeof(int), hipMemcpyDeviceToHost, stream), "read flags");
    BK_CHECK(hipMemcpyAsync(&root, dNodes, sizeof root, hipMemcpyDeviceToHost, stream), "read root");
    BK_CHECK(hipStreamSynchronize(stream), "sync");
    if (flags & 2) {
        *error = "device bake: instance primitives are baked on the host (nnbvh_scene_create_instanced)";
        return false;
    }
    if ((flags & 16) && !d_normals_) {
        *error = "scene_create: smooth alpha-tested primitives need the vertex normals "
                 "(nnbvh_scene_create_with_normals)";
        return false;
    }
    if ((flags & 32) && !d_prim_alpha_) {
        *error = "scene_create: NNBVH_PRIM_ALPHA_PATCH primitives need the per-primitive alpha array "
                 "(nnbvh_scene_create_with_attributes)";
        return false;
    }
    if ((flags & 64) && !d_uvs_) {
        *error = "scene_create: NNBVH_PRIM_ALPHA_PATCH_UV primitives need the per-vertex (u, v) array "
                 "(nnbvh_scene_create_with_attributes)";
        return false;
    }
    if (nSlots <= 0 || nSlots >= 0x7ffffffe) {
        *error = "device bake: primitive stream exceeds 2^31 slots";
        return false;
    }
    // ONE allocation: the interior records, then (256-B aligned) the primitive stream and 64 B of padding —
    // the lean traversal kernels reach both with a 32-bit offset from one scalar base, and a lane that
    // fetches a primitive through the interior record's 64-B load sequence may read one slot past it
    void *dWide = nullptr, *dStream = nullptr;
    const size_t wideBytes = (((size_t)std::max(nInterior, 1) * sizeof(WideNode)) + 255) & ~(size_t)255;
    BK_CHECK(hipMalloc(&dWide, wideBytes + (size_t)nSlots * 16 + 64), "hipMalloc(nodes + primitives)");
    dStream = (char *)dWide + wideBytes;
    BK_CHECK(hipMemsetAsync((char *)dStream + (size_t)nSlots * 16, 0, 64, stream), "memset");
    hipLaunchKernelGGL(k_bake_stream, dim3(gp), dim3(kBk), 0, stream, dPrims, n_prims, dVerts, (const float *)d_normals_, (const float *)d_uvs_, (const float *)d_prim_alpha_, dSlotOf, dLeafLast,
                       (float4 *)dStream);
    hipLaunchKernelGGL(k_bake_wide, dim3(gn), dim3(kBk), 0, stream, dNodes, n_nodes, dOrd, dSlotOf, (float4 *)dWide);
    int rootSlot = 0;
    if (root.nprims != 0)
        (void)hipMemcpyAsync(&rootSlot, dSlotOf + root.offset, sizeof(int), hipMemcpyDeviceToHost, stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        (void)hipFree(dWide);
        *error = std::string("device bake: ") + hipGetErrorString(e);
        return false;
    }
    out->d_wide = dWide;
    out->d_prims = dStream;
    out->n_interior = nInterior;
    out->n_slots = nSlots;
    out->root_ref = root.nprims == 0 ? 0 : ~rootSlot;  // interior root = record 0
    std::memcpy(out->bounds, root.pmin, 12);
    std::memcpy(out->bounds + 3, root.pmax, 12);
    out->has_host_prims = flags & 1;
    out->has_patches = (flags & 4) ? 1 : 0;
    out->has_alpha = (flags & 32) ? 2 : ((flags & 8) ? 1 : 0);  // 2: alpha-tested PATCHES present (the ALPHA = 2 kernels)
    return true;
}

bool gather_prim_alpha_on_device(void *d_ordered, int n_prims, const float *prim_alpha, const int32_t *caller_ids,
                                 void **d_alpha_out, std::string *error) {
    void *dIn = nullptr, *dIds = nullptr, *dOut = nullptr;
    auto fail = [&](const char *what, hipError_t e) {
        for (void *p : {dIn, dIds, dOut})
            if (p) (void)hipFree(p);
        *error = std::string("device bake: ") + what + ": " + hipGetErrorString(e);
        return false;
    };
    const size_t bytes = (size_t)n_prims * 4;
    hipError_t e = hipMalloc(&dIn, bytes);
    if (e == hipSuccess) e = hipMalloc(&dIds, bytes);
    if (e == hipSuccess) e = hipMalloc(&dOut, bytes);
    if (e != hipSuccess) return fail("hipMalloc(primitive alpha)", e);
    e = hipMemcpy(dIn, prim_alpha, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dIds, caller_ids, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail("hipMemcpy(primitive alpha)", e);
    hipLaunchKernelGGL(k_gather_alpha_restore_ids, dim3((n_prims + kBk - 1) / kBk), dim3(kBk), 0, nullptr,
                       (nnbvh_prim *)d_ordered, n_prims, (const float *)dIn, (const int *)dIds, (float *)dOut);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail("gather", e);
    (void)hipFree(dIn);
    (void)hipFree(dIds);
    *d_alpha_out = dOut;
    return true;
}

}  // namespace nnbvh
