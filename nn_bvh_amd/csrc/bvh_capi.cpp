// bvh_capi.cpp — the C ABI of include/nnbvh.h: tree validation, baking of the device
// layout, workspaces and kernel launches.  Host code only (compiled with hipcc for the
// HIP runtime API).  There is deliberately no CPU traversal in this library: if no HIP
// device is usable the intersect entry points fail with NNBVH_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh_trace.h"
#include "bvh_build_gpu.h"
#include "interaction.h"
#include "wavefront.h"
#include "wavefront2.h"

namespace nnbvh {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

static bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hip_ok(hipSetDevice(dev), "hipSetDevice");
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct Workspace {
    unsigned *queue = nullptr;  // kMaxQueues heads
    uint2 *spill = nullptr;
    // grow-only staging for the host-buffer entry points
    void *d_in = nullptr, *d_out = nullptr, *d_aux0 = nullptr, *d_aux1 = nullptr;
    size_t in_bytes = 0, out_bytes = 0, aux_bytes = 0;
    // grow-only scratch of the multi-pass entry points (IntersectShadowTr / IntersectOneRandom): ping-pong ray and
    // hit buffers, per-item state, counters.  Per stream like the queue heads: the calls are asynchronous on their
    // stream, so two streams must not share them
    static constexpr int kScratch = 12;
    void *scratch[kScratch] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[kScratch] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// one slot of the host-buffer pipeline (nnbvh_intersect_closest / _any): device chunk buffers, events and — for
// callers with pageable memory — pinned staging buffers
struct HostSlot {
    hipEvent_t ev_in = nullptr, ev_traced = nullptr, ev_out = nullptr;  // rays uploaded / chunk traced / results down
    void *d_in = nullptr, *h_in = nullptr;
    size_t d_in_bytes = 0, h_in_bytes = 0;
    void *d_out[3] = {nullptr, nullptr, nullptr}, *h_out[3] = {nullptr, nullptr, nullptr};
    size_t d_out_bytes[3] = {0, 0, 0}, h_out_bytes[3] = {0, 0, 0};
};

}  // namespace nnbvh

using namespace nnbvh;

struct nnbvh_scene {
    int device = 0;
    int n_cus = 0;
    int n_interior = 0;
    int64_t n_slots = 0;
    int depth = 0;
    float bounds[6];
    int root_ref = 0;
    float4 *d_wide = nullptr;
    float4 *d_prims = nullptr;
    float *d_anim = nullptr;  // AnimatedPrimitive table (kAnimStride floats per instance), or null
    size_t device_bytes = 0;
    // tuning (speed only)
    int window = 8;
    int blocks_per_cu = 0;  // 0 = from the occupancy query
    int xcd_queues = 1;
    int prim_weight = 32;
    int refill_weight = 8;
    unsigned long long *d_stats = nullptr;  // diagnostics (NNBVH_STATS builds)
    int instanced = 0;      // two-level scene: use the INST kernels
    int has_host_prims = 0;
    int has_patches = 1;    // 0: no bilinear patches, the lean kernels (no ray direction parked in LDS) run
    int has_alpha = 0;      // 1: alpha-tested triangles present (the ALPHA kernels run), 2: alpha-tested patches too
    int fused_batches = 1;  // nnbvh_trace_batches_device: one mode-3 launch where the batches allow it
    int int_repeat = 3;
    int prim_repeat = 2;
    int prim_min = 8;  // only read by builds with -DNNBVH_MERGED=1  // merged trips of the lean kernels (bvh_trace.hip); 0 = separate interior / primitive trips
    int max_grid_threads = 0;
    double build_ms[1] = {0};  // device build time of nnbvh_scene_create_gpu_build
    std::mutex mu;
    std::map<hipStream_t, Workspace> workspaces;
    static constexpr int kHostSlots = 3, kHostChunks = 6;
    HostSlot host_slots[kHostSlots];
    hipStream_t host_up = nullptr, host_trace = nullptr, host_down = nullptr;
    int64_t host_chunk = 1 << 20;  // least rays per chunk of the host-buffer pipeline (at most kHostChunks chunks)
    // fork/join machinery of nnbvh_trace_batches_device
    static constexpr int kSideStreams = 4;
    hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
};

// -------------------------------------------------------------------------------------
// Tree validation: everything the kernels index with is range-checked here once, so a
// malformed tree is an NNBVH_ERR_ARG at create time, never a device fault.
// Validates the DFS-laid-out tree occupying nodes[root, root + n_tree).  `covered` / `leaf_last`
// are shared by all trees of a scene (top level + instanced children).
static bool validate_tree(const nnbvh_linear_node *nodes, int root, int n_tree, const nnbvh_prim *prims,
                          int n_prims, bool allow_instances, std::vector<uint8_t> &covered,
                          std::vector<uint8_t> &leaf_last, int *depth_out) {
    if (n_tree < 1) {
        set_error("scene_create: tree has no nodes");
        return false;
    }
    const int n_nodes = root + n_tree;
    // iterative DFS; each frame = (node, end bound of its subtree range, depth)
    struct Frame {
        int node, end, depth;
    };
    std::vector<Frame> st;
    st.push_back({root, n_nodes, 0});
    int visited = 0, max_depth = 0;
    while (!st.empty()) {
        Frame f = st.back();
        st.pop_back();
        ++visited;
        if (f.node < 0 || f.node >= f.end) {
            set_error("scene_create: node index outside its subtree range");
            return false;
        }
        max_depth = std::max(max_depth, f.depth);
        const nnbvh_linear_node &nd = nodes[f.node];
        // Bounds3f with min <= max on every axis (and no NaN): what every builder emits, and what the
        // interior step's form of the slab test (trace_math.h slab_entry_key) is equal to the
        // reference's for
        for (int k = 0; k < 3; ++k)
            if (!(nd.pmin[k] <= nd.pmax[k])) {
                set_error("scene_create: node bounds with min > max (or NaN)");
                return false;
            }
        if (nd.nprims > 0) {
            if (f.node + 1 != f.end) {
                set_error("scene_create: leaf does not close its subtree range (not a DFS layout)");
                return false;
            }
            if (nd.offset < 0 || (int64_t)nd.offset + nd.nprims > n_prims) {
                set_error("scene_create: leaf primitive range out of bounds");
                return false;
            }
            for (int i = 0; i < nd.nprims; ++i) {
                if (covered[(size_t)nd.offset + i]) {
                    set_error("scene_create: primitive referenced by two leaves");
                    return false;
                }
                covered[(size_t)nd.offset + i] = 1;
                if (!allow_instances && prims[(size_t)nd.offset + i].kind == NNBVH_PRIM_INSTANCE) {
                    set_error("scene_create: nested instances are not supported");
                    return false;
                }
            }
            leaf_last[(size_t)nd.offset + nd.nprims - 1] = 1;
        } else {
            if (nd.axis > 2) {
                set_error("scene_create: interior node axis > 2");
                return false;
            }
            const int c0 = f.node + 1, c1 = nd.offset;
            if (c1 <= c0 || c1 >= f.end) {
                set_error("scene_create: secondChildOffset outside the node's subtree range");
                return false;
            }
            st.push_back({c1, f.end, f.depth + 1});
            st.push_back({c0, c1, f.depth + 1});
        }
    }
    if (visited != n_tree) {
        set_error("scene_create: unreachable nodes in the array");
        return false;
    }
    *depth_out = max_depth;
    return true;
}

// The reference's degenerate-triangle test (shapes.cpp:176-177) with its exact float32
// arithmetic: Cross via DifferenceOfProducts (util/vecmath.h:999-1004, util/math.h:569-575,
// fmaf where the reference has FMA), LengthSquared as x*x + y*y + z*z (vecmath.h:948-950).
// This translation unit is compiled with -ffp-contract=off.
static float dop_host(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = std::fma(a, b, -cd);
    float err = std::fma(-c, d, cd);
    return diff + err;
}
static bool triangle_is_degenerate(const float *p0, const float *p1, const float *p2) {
    float v[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    float w[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    float cx = dop_host(v[1], w[2], v[2], w[1]);
    float cy = dop_host(v[2], w[0], v[0], w[2]);
    float cz = dop_host(v[0], w[1], v[1], w[0]);
    return cx * cx + cy * cy + cz * cz == 0.0f;
}

static void put3(float *q, int at, const float *v) {
    q[at] = v[0];
    q[at + 1] = v[1];
    q[at + 2] = v[2];
}

// one entry of the device's animation table (layout: anim_math.h)
static float quat_angle_between_host(const float a[4], const float b[4]);
static float sin_x_over_x_host(float x);
static void fill_anim_entry(const nnbvh_animated_transform &a, float *t) {
    std::memcpy(t, a.T, 24);
    std::memcpy(t + 6, a.R, 32);
    std::memcpy(t + 14, a.S, 128);
    t[46] = a.start_time;
    t[47] = a.end_time;
    t[48] = quat_angle_between_host(a.R[0], a.R[1]);
    t[49] = sin_x_over_x_host(t[48]);
    std::memcpy(t + 50, a.start_inv, 48);
    std::memcpy(t + 62, a.end_inv, 48);
    t[74] = a.actually_animated ? 1.0f : 0.0f;
}

extern "C" {

const char *nnbvh_last_error(void) { return g_error.c_str(); }

int nnbvh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// a scene object around arrays baked on the device (bvh_bake.hip)
static nnbvh_scene *scene_from_baked(const BakedScene &b, int depth, int device) {
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) {
        (void)hipFree(b.d_wide);  // one allocation: d_prims points into it
        return nullptr;
    }
    auto *s = new nnbvh_scene;
    s->device = device;
    s->n_cus = prop.multiProcessorCount;
    s->n_interior = b.n_interior;
    s->n_slots = b.n_slots;
    s->depth = depth;
    std::memcpy(s->bounds, b.bounds, sizeof b.bounds);
    s->root_ref = b.root_ref;
    s->instanced = 0;
    s->has_host_prims = (b.has_host_prims || b.has_alpha) ? 1 : 0;
    s->has_patches = (b.has_patches || b.has_alpha) ? 1 : 0;  // the alpha test hashes the ray direction, parked with the patches' one
    s->has_alpha = b.has_alpha;
    s->max_grid_threads = s->n_cus * 8 * kBlockThreads;
    s->d_wide = (float4 *)b.d_wide;
    s->d_prims = (float4 *)b.d_prims;
    s->device_bytes = (size_t)std::max(b.n_interior, 1) * sizeof(WideNode) +
                      std::max<size_t>((size_t)b.n_slots, 1) * 16;
    if (const char *e = std::getenv("NNBVH_LAYOUT")) {  // memory order of records / leaves (bvh_layout.cpp)
        const char *t = std::getenv("NNBVH_LAYOUT_TOP");
        std::string err;
        if (!relayout_scene(atoi(e), &s->d_wide, &s->d_prims, &s->n_interior, &s->n_slots, &s->root_ref,
                            t ? atoi(t) : 12, &err)) {
            set_error(err);
            (void)hipFree(s->d_wide);
            delete s;
            return nullptr;
        }
        s->device_bytes = (size_t)std::max(s->n_interior, 1) * sizeof(WideNode) + (size_t)s->n_slots * 16;
    }
    if (hipMalloc((void **)&s->d_stats, 16 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(s->d_stats, 0, 16 * sizeof(unsigned long long));
    if (const char *e = std::getenv("NNBVH_STACK_WINDOW")) nnbvh_scene_set_option(s, "stack_window", atoi(e));
    if (const char *e = std::getenv("NNBVH_BLOCKS_PER_CU")) nnbvh_scene_set_option(s, "blocks_per_cu", atoi(e));
    if (const char *e = std::getenv("NNBVH_XCD_QUEUES")) nnbvh_scene_set_option(s, "xcd_queues", atoi(e));
    if (const char *e = std::getenv("NNBVH_REFILL_WEIGHT")) nnbvh_scene_set_option(s, "refill_weight", atoi(e));
    if (const char *e = std::getenv("NNBVH_PRIM_WEIGHT")) nnbvh_scene_set_option(s, "prim_weight", atoi(e));
    if (const char *e = std::getenv("NNBVH_PRIM_MIN")) nnbvh_scene_set_option(s, "prim_min", atoi(e));
    return s;
}

// Single-level scenes: the validated tree is uploaded as it is and baked on the device (the same
// arrays the host code below produces for two-level scenes, without the host pass over every node
// and primitive).
static nnbvh_scene *create_scene_device_bake(const nnbvh_linear_node *nodes, int n_nodes, const nnbvh_prim *prims,
                                             int n_prims, const float *verts, int n_verts, int depth, int device,
                                             const float *normals, const float *prim_alpha, const float *uvs) {
    int n_dev = nnbvh_device_count();
    if (n_dev <= 0 || device < 0 || device >= n_dev) {
        set_error("scene_create: no usable HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return nullptr;
    void *d_nodes = nullptr, *d_prims = nullptr, *d_verts = nullptr, *d_normals = nullptr, *d_alpha = nullptr,
         *d_uvs = nullptr;
    const size_t nb = (size_t)n_nodes * sizeof(nnbvh_linear_node), pbytes = (size_t)n_prims * sizeof(nnbvh_prim),
                 vb = (size_t)n_verts * 12;
    BakedScene b;
    std::string err;
    bool ok = hip_ok(hipMalloc(&d_nodes, nb), "hipMalloc(tree)") && hip_ok(hipMalloc(&d_prims, pbytes), "hipMalloc(primitives)") &&
              hip_ok(hipMalloc(&d_verts, vb), "hipMalloc(vertices)") &&
              hip_ok(hipMemcpy(d_nodes, nodes, nb, hipMemcpyHostToDevice), "hipMemcpy(tree)") &&
              hip_ok(hipMemcpy(d_prims, prims, pbytes, hipMemcpyHostToDevice), "hipMemcpy(primitives)") &&
              hip_ok(hipMemcpy(d_verts, verts, vb, hipMemcpyHostToDevice), "hipMemcpy(vertices)");
    if (ok && normals)
        ok = hip_ok(hipMalloc(&d_normals, vb), "hipMalloc(normals)") &&
             hip_ok(hipMemcpy(d_normals, normals, vb, hipMemcpyHostToDevice), "hipMemcpy(normals)");
    if (ok && prim_alpha)
        ok = hip_ok(hipMalloc(&d_alpha, (size_t)n_prims * 4), "hipMalloc(primitive alpha)") &&
             hip_ok(hipMemcpy(d_alpha, prim_alpha, (size_t)n_prims * 4, hipMemcpyHostToDevice), "hipMemcpy(primitive alpha)");
    if (ok && uvs)
        ok = hip_ok(hipMalloc(&d_uvs, (size_t)n_verts * 8), "hipMalloc(uvs)") &&
             hip_ok(hipMemcpy(d_uvs, uvs, (size_t)n_verts * 8, hipMemcpyHostToDevice), "hipMemcpy(uvs)");
    if (ok && !bake_on_device(d_nodes, n_nodes, d_prims, n_prims, d_verts, device, &b, &err, d_normals, d_alpha, d_uvs)) {
        set_error(err);
        ok = false;
    }
    for (void *p : {d_nodes, d_prims, d_verts, d_normals, d_alpha, d_uvs})
        if (p) (void)hipFree(p);
    return ok ? scene_from_baked(b, depth, device) : nullptr;
}

// AngleBetween(Quaternion, Quaternion) (util/vecmath.h:1138-1143) and SinXOverX (util/math.h:340-344) with
// the host's libm, as the reference's Slerp evaluates them; they depend on the transform only
static float quat_dot_host(const float a[4], const float b[4]) {
    return (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) + a[3] * b[3];
}
static float quat_angle_between_host(const float q1[4], const float q2[4]) {
    float t[4];
    const bool neg = quat_dot_host(q1, q2) < 0;
    for (int k = 0; k < 4; ++k) t[k] = neg ? q1[k] + q2[k] : q2[k] - q1[k];
    float x = std::sqrt(quat_dot_host(t, t)) / 2;
    x = x < -1 ? -1 : (x > 1 ? 1 : x);
    return neg ? 3.14159265358979323846f - 2 * std::asin(x) : 2 * std::asin(x);
}
// launch_trace's `patches`: bit 0 patches (or alpha: the parked ray direction), bit 1 alpha-tested triangles,
// bit 2 alpha-tested patches
static int patch_bits(const nnbvh_scene *s) { return s->has_patches + 2 * (s->has_alpha != 0) + 4 * (s->has_alpha == 2); }

static float sin_x_over_x_host(float x) {
    if (1 - x * x == 1) return 1;
    return std::sin(x) / x;
}

static nnbvh_scene *create_scene(const nnbvh_linear_node *nodes, int n_nodes, int n_top_nodes,
                                 const nnbvh_prim *prims, int n_prims, const float *verts,
                                 int n_verts, const nnbvh_instance *instances, int n_instances,
                                 int device, const nnbvh_animated_transform *animated = nullptr,
                                 const float *normals = nullptr, const float *prim_alpha = nullptr,
                                 const float *uvs = nullptr) {
    if (!nodes || !prims || !verts || n_prims <= 0 || n_verts <= 0 || n_instances < 0 ||
        (n_instances > 0 && !instances) || n_top_nodes < 1 || n_top_nodes > n_nodes) {
        set_error("scene_create: null or empty input array");
        return nullptr;
    }
    std::vector<uint8_t> leaf_last((size_t)n_prims, 0), covered((size_t)n_prims, 0);
    int depth = 0;
    if (!validate_tree(nodes, 0, n_top_nodes, prims, n_prims, n_instances > 0, covered, leaf_last,
                       &depth))
        return nullptr;
    // instanced children: trees in nodes[n_top_nodes, n_nodes); several instances may share one
    int child_depth = 0;
    std::vector<uint8_t> tree_seen((size_t)n_nodes, 0);
    for (int k = 0; k < n_instances; ++k) {
        const nnbvh_instance &in = instances[k];
        if (in.root < n_top_nodes || in.n_nodes < 1 || (int64_t)in.root + in.n_nodes > n_nodes) {
            set_error("scene_create: instance child tree outside the node array");
            return nullptr;
        }
        if (tree_seen[(size_t)in.root]) continue;
        tree_seen[(size_t)in.root] = 1;
        int d = 0;
        if (!validate_tree(nodes, in.root, in.n_nodes, prims, n_prims, false, covered, leaf_last, &d))
            return nullptr;
        child_depth = std::max(child_depth, d + 1);
    }
    depth += child_depth;  // pending entries of the outer walk + those of the child walk
    if (depth > kMaxStack) {
        // the reference's nodesToVisit[64] (aggregates.cpp:538) would overflow silently
        set_error("scene_create: tree deeper than the 64-entry traversal stack");
        return nullptr;
    }
    // prim stream
    std::vector<int64_t> slot_of((size_t)n_prims + 1, 0);
    for (int k = 0; k < n_prims; ++k) {
        const nnbvh_prim &p = prims[k];
        int nv, nslots;
        if (is_smooth_alpha_kind(p.kind)) {
            nv = 3;
            nslots = 6;
            if (!normals) {
                set_error("scene_create: NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH primitives need the vertex normals "
                          "(nnbvh_scene_create_with_normals)");
                return nullptr;
            }
        } else if (is_alpha_patch_kind(p.kind)) {
            nv = 4;
            nslots = alpha_patch_slots(p.kind);
            if (!prim_alpha || (is_smooth_alpha_patch_kind(p.kind) && !normals) || (is_uv_alpha_patch_kind(p.kind) && !uvs)) {
                set_error("scene_create: NNBVH_PRIM_ALPHA_PATCH primitives need the per-primitive alpha array, the "
                          "smooth ones the vertex normals, the _UV ones the vertex uvs too "
                          "(nnbvh_scene_create_with_attributes)");
                return nullptr;
            }
        } else if (is_triangle_kind(p.kind)) nv = nslots = 3;
        else if (p.kind == NNBVH_PRIM_BILINEAR_PATCH) nv = nslots = 4;
        else if (p.kind == NNBVH_PRIM_HOST) {
            nv = 0;
            nslots = 3;
        } else if (p.kind == NNBVH_PRIM_INSTANCE) {
            nv = 0;
            nslots = 6;
            if (p.v[0] < 0 || p.v[0] >= n_instances) {
                set_error("scene_create: instance index out of range");
                return nullptr;
            }
        } else {
            set_error("scene_create: unknown primitive kind");
            return nullptr;
        }
        for (int j = 0; j < nv; ++j)
            if (p.v[j] < 0 || p.v[j] >= n_verts) {
                set_error("scene_create: vertex index out of range");
                return nullptr;
            }
        slot_of[(size_t)k + 1] = slot_of[(size_t)k] + nslots;
    }
    const int64_t n_slots = slot_of[(size_t)n_prims];
    if (n_slots >= 0x7ffffffeLL) {
        set_error("scene_create: primitive stream exceeds 2^31 slots");
        return nullptr;
    }
    if (n_instances == 0)
        return create_scene_device_bake(nodes, n_nodes, prims, n_prims, verts, n_verts, depth, device, normals, prim_alpha, uvs);
    // interior record numbers (global over all trees) and node refs
    std::vector<int> ord((size_t)n_nodes, -1);
    int n_interior = 0;
    for (int i = 0; i < n_nodes; ++i)
        if (nodes[i].nprims == 0) ord[(size_t)i] = n_interior++;
    auto ref_of = [&](int i) -> int32_t {
        return nodes[i].nprims == 0 ? ord[(size_t)i]
                                    : (int32_t) ~(uint32_t)slot_of[(size_t)nodes[i].offset];
    };
    std::vector<float> stream((size_t)n_slots * 4, 0.0f);
    bool has_host = false, has_alpha = false, has_alpha_patch = false;
    for (int k = 0; k < n_prims; ++k) {
        const nnbvh_prim &p = prims[k];
        float *s = &stream[(size_t)slot_of[(size_t)k] * 4];
        uint32_t flags = leaf_last[(size_t)k] ? kPrimLast : 0u;
        if (p.kind == NNBVH_PRIM_INSTANCE) {
            const nnbvh_instance &in = instances[p.v[0]];
            const nnbvh_linear_node &root = nodes[in.root];
            flags |= kPrimInstance;
            if (animated && animated[p.v[0]].actually_animated) flags |= kPrimAnimated;
            put3(s, 0, root.pmin);
            put3(s, 4, root.pmax);
            std::memcpy(&s[3], &p.v[0], 4);  // instance index (reported +1 in nnbvh_hit.instance)
            std::memcpy(&s[7], &flags, 4);
            std::memcpy(&s[8], in.prim_from_render, 48);
            const int32_t rref = ref_of(in.root);
            std::memcpy(&s[20], &rref, 4);
            continue;
        }
        if (p.kind == NNBVH_PRIM_HOST) {
            flags |= kPrimHost;
            has_host = true;
            std::memcpy(&s[3], &p.id, 4);
            std::memcpy(&s[7], &flags, 4);
            continue;
        }
        const int nv = is_triangle_kind(p.kind) ? 3 : 4;
        for (int j = 0; j < nv; ++j) put3(s, 4 * j, verts + 3 * (size_t)p.v[j]);
        if (p.kind == NNBVH_PRIM_BILINEAR_PATCH) flags |= kPrimPatch;
        if (is_alpha_patch_kind(p.kind)) {  // as k_bake_stream lays them out (bvh_bake.hip)
            flags |= kPrimPatch | kPrimAlpha | (is_flipped_alpha_patch_kind(p.kind) ? kPrimFlipN : 0u);
            s[11] = prim_alpha[k];
            has_alpha = has_alpha_patch = true;
            if (is_smooth_alpha_patch_kind(p.kind)) {
                flags |= kPrimSmooth;
                for (int j = 0; j < 4; ++j) put3(s, 16 + 4 * j, normals + 3 * (size_t)p.v[j]);
            }
            if (is_uv_alpha_patch_kind(p.kind)) {
                flags |= kPrimUV;
                float *u = s + (is_smooth_alpha_patch_kind(p.kind) ? 32 : 16);
                for (int j = 0; j < 4; ++j) {
                    u[2 * j] = uvs[2 * (size_t)p.v[j]];
                    u[2 * j + 1] = uvs[2 * (size_t)p.v[j] + 1];
                }
            }
        }
        if (is_flat_alpha_kind(p.kind) || is_smooth_alpha_kind(p.kind)) {
            flags |= kPrimAlpha;
            if (p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED || p.kind == NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED) flags |= kPrimFlipN;
            std::memcpy(&s[11], &p.v[3], 4);  // alpha (float bit pattern) in slot 2's fourth word
            has_alpha = true;
            if (is_smooth_alpha_kind(p.kind)) {  // the three vertex normals in slots 3..5
                flags |= kPrimSmooth;
                for (int j = 0; j < 3; ++j) put3(s, 12 + 4 * j, normals + 3 * (size_t)p.v[j]);
            }
        }
        if (is_triangle_kind(p.kind) &&
            triangle_is_degenerate(verts + 3 * (size_t)p.v[0], verts + 3 * (size_t)p.v[1],
                                   verts + 3 * (size_t)p.v[2]))
            flags |= kPrimDegenerate;
        std::memcpy(&s[3], &p.id, 4);
        std::memcpy(&s[7], &flags, 4);
    }
    // interior records
    std::vector<WideNode> wide((size_t)std::max(n_interior, 1));
    std::memset(wide.data(), 0, wide.size() * sizeof(WideNode));
    for (int i = 0; i < n_nodes; ++i) {
        if (nodes[i].nprims != 0) continue;
        WideNode &w = wide[(size_t)ord[(size_t)i]];
        const nnbvh_linear_node &c0 = nodes[i + 1], &c1 = nodes[nodes[i].offset];
        put3(w.q, 0, c0.pmin);
        put3(w.q, 3, c0.pmax);
        put3(w.q, 6, c1.pmin);
        put3(w.q, 9, c1.pmax);
        w.ref0 = ref_of(i + 1);
        w.ref1 = ref_of(nodes[i].offset);
        w.axis = nodes[i].axis;
        w.pad = 0;
    }

    int n_dev = nnbvh_device_count();
    if (n_dev <= 0 || device < 0 || device >= n_dev) {
        set_error("scene_create: no usable HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return nullptr;
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return nullptr;

    auto *s = new nnbvh_scene;
    s->device = device;
    s->n_cus = prop.multiProcessorCount;
    s->n_interior = n_interior;
    s->n_slots = n_slots;
    s->depth = depth;
    std::memcpy(s->bounds, nodes[0].pmin, 12);
    std::memcpy(s->bounds + 3, nodes[0].pmax, 12);
    s->root_ref = ref_of(0);
    s->instanced = n_instances > 0 ? 1 : 0;
    s->has_host_prims = (has_host || has_alpha) ? 1 : 0;  // an alpha re-trace that hits voids the ray like a host primitive
    s->has_alpha = has_alpha_patch ? 2 : (has_alpha ? 1 : 0);
    s->has_patches = 1;  // two-level scenes always run the general kernels
    s->max_grid_threads = s->n_cus * 8 * kBlockThreads;
    // one allocation, as bake_on_device makes it: records, then the 256-B aligned primitive stream + 64 B
    const size_t wide_bytes = (wide.size() * sizeof(WideNode) + 255) & ~(size_t)255;
    const size_t prim_bytes = std::max<size_t>((size_t)n_slots, 1) * 16 + 64;
    bool ok = hip_ok(hipMalloc((void **)&s->d_wide, wide_bytes + prim_bytes), "hipMalloc(nodes + prims)");
    if (ok) s->d_prims = (float4 *)((char *)s->d_wide + wide_bytes);
    ok = ok && hip_ok(hipMemset((char *)s->d_prims + prim_bytes - 64, 0, 64), "hipMemset(pad)") &&
         hip_ok(hipMemcpy(s->d_wide, wide.data(), wide.size() * sizeof(WideNode), hipMemcpyHostToDevice),
                "hipMemcpy(nodes)") &&
         hip_ok(hipMemcpy(s->d_prims, stream.data(), (size_t)n_slots * 16, hipMemcpyHostToDevice),
                "hipMemcpy(prims)");
    if (!ok) {
        if (s->d_wide) (void)hipFree(s->d_wide);
        delete s;
        return nullptr;
    }
    s->device_bytes = wide_bytes + prim_bytes;
    if (animated && n_instances > 0) {
        std::vector<float> table((size_t)n_instances * kAnimStride, 0.0f);
        for (int k = 0; k < n_instances; ++k) fill_anim_entry(animated[k], &table[(size_t)k * kAnimStride]);
        if (!hip_ok(hipMalloc((void **)&s->d_anim, table.size() * 4), "hipMalloc(animation table)") ||
            !hip_ok(hipMemcpy(s->d_anim, table.data(), table.size() * 4, hipMemcpyHostToDevice),
                    "hipMemcpy(animation table)")) {
            nnbvh_scene_destroy(s);
            return nullptr;
        }
    }
    if (hipMalloc((void **)&s->d_stats, 16 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(s->d_stats, 0, 16 * sizeof(unsigned long long));
    if (const char *e = std::getenv("NNBVH_STACK_WINDOW")) nnbvh_scene_set_option(s, "stack_window", atoi(e));
    if (const char *e = std::getenv("NNBVH_BLOCKS_PER_CU")) nnbvh_scene_set_option(s, "blocks_per_cu", atoi(e));
    if (const char *e = std::getenv("NNBVH_XCD_QUEUES")) nnbvh_scene_set_option(s, "xcd_queues", atoi(e));
    if (const char *e = std::getenv("NNBVH_REFILL_WEIGHT")) nnbvh_scene_set_option(s, "refill_weight", atoi(e));
    if (const char *e = std::getenv("NNBVH_PRIM_WEIGHT")) nnbvh_scene_set_option(s, "prim_weight", atoi(e));
    if (const char *e = std::getenv("NNBVH_PRIM_MIN")) nnbvh_scene_set_option(s, "prim_min", atoi(e));
    return s;
}

// Triangles in, traceable scene out, without the tree leaving the device: device build
// (bvh_build_gpu.hip) + device bake (bvh_bake.hip).  Same tree, same baked arrays, hence the same
// results as nnbvh_build_create + nnbvh_scene_create.
nnbvh_scene *nnbvh_scene_create_gpu_build(const nnbvh_prim *prims, int n_prims, const float *verts,
                                          int n_verts, const float *prim_bounds,
                                          int max_prims_in_node, int split_method, int device) {
    return nnbvh_scene_create_gpu_build_with_attributes(prims, n_prims, verts, n_verts, prim_bounds, nullptr, nullptr,
                                                        nullptr, max_prims_in_node, split_method, device);
}

nnbvh_scene *nnbvh_scene_create_gpu_build_with_attributes(const nnbvh_prim *prims_in, int n_prims, const float *verts,
                                                          int n_verts, const float *prim_bounds, const float *normals,
                                                          const float *uvs, const float *prim_alpha,
                                                          int max_prims_in_node, int split_method, int device) {
    const nnbvh_prim *prims = prims_in;
    // with a per-primitive array to carry along, the build runs with ids = positions; the bake's gather pass puts the
    // caller's ids back (bvh_bake.hip)
    std::vector<nnbvh_prim> tagged;
    std::vector<int32_t> caller_ids;
    if (prims_in && prim_alpha && n_prims > 0) {
        tagged.assign(prims_in, prims_in + n_prims);
        caller_ids.resize((size_t)n_prims);
        for (int i = 0; i < n_prims; ++i) {
            caller_ids[(size_t)i] = tagged[(size_t)i].id;
            tagged[(size_t)i].id = i;
        }
        prims = tagged.data();
    }
    if (!prims || !verts || n_prims <= 0 || n_verts <= 0) {
        set_error("scene_create_gpu_build: empty primitive or vertex array");
        return nullptr;
    }
    if (split_method != NNBVH_SPLIT_SAH && split_method != NNBVH_SPLIT_HLBVH) {
        set_error("scene_create_gpu_build: only the sah and hlbvh split methods are built on the device");
        return nullptr;
    }
    int n_dev = nnbvh_device_count();
    if (n_dev <= 0 || device < 0 || device >= n_dev) {
        set_error("scene_create_gpu_build: no usable HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return nullptr;
    GpuBuildResult r;
    r.keep_on_device = true;
    std::string err;
    const bool built = split_method == NNBVH_SPLIT_SAH
                           ? gpu_sah(prims, n_prims, verts, n_verts, prim_bounds, max_prims_in_node, device, &r, &err)
                           : gpu_hlbvh(prims, n_prims, verts, n_verts, prim_bounds, max_prims_in_node, device, &r, &err);
    if (!built) {
        set_error(err);
        return nullptr;
    }
    BakedScene b;
    bool ok = r.depth <= kMaxStack;
    if (!ok) err = "scene_create: tree deeper than the 64-entry traversal stack";
    void *d_normals = nullptr, *d_alpha = nullptr, *d_uvs = nullptr;
    if (ok && uvs) {
        ok = hip_ok(hipMalloc(&d_uvs, (size_t)n_verts * 8), "hipMalloc(uvs)") &&
             hip_ok(hipMemcpy(d_uvs, uvs, (size_t)n_verts * 8, hipMemcpyHostToDevice), "hipMemcpy(uvs)");
        if (!ok) err = nnbvh_last_error();
    }
    if (ok && normals) {
        ok = hip_ok(hipMalloc(&d_normals, (size_t)n_verts * 12), "hipMalloc(normals)") &&
             hip_ok(hipMemcpy(d_normals, normals, (size_t)n_verts * 12, hipMemcpyHostToDevice), "hipMemcpy(normals)");
        if (!ok) err = nnbvh_last_error();
    }
    if (ok && prim_alpha)
        ok = gather_prim_alpha_on_device(r.d_ordered, n_prims, prim_alpha, caller_ids.data(), &d_alpha, &err);
    ok = ok && bake_on_device(r.d_nodes, r.total_nodes, r.d_ordered, n_prims, r.d_verts, device, &b, &err, d_normals, d_alpha,
                              d_uvs);
    for (void *p : {r.d_nodes, r.d_ordered, r.d_verts, d_normals, d_alpha, d_uvs})
        if (p) (void)hipFree(p);
    if (!ok) {
        set_error(err);
        return nullptr;
    }
    nnbvh_scene *s = scene_from_baked(b, r.depth, device);
    if (!s) return nullptr;
    s->build_ms[0] = r.ms[0] + r.ms[1] + r.ms[2] + r.ms[3] + r.ms[4];
    return s;
}

nnbvh_scene *nnbvh_scene_create(const nnbvh_linear_node *nodes, int n_nodes,
                                const nnbvh_prim *prims, int n_prims, const float *verts,
                                int n_verts, int device) {
    return create_scene(nodes, n_nodes, n_nodes, prims, n_prims, verts, n_verts, nullptr, 0, device);
}

nnbvh_scene *nnbvh_scene_create_with_attributes(const nnbvh_linear_node *nodes, int n_nodes, const nnbvh_prim *prims,
                                                int n_prims, const float *verts, const float *normals,
                                                const float *uvs, const float *prim_alpha, int n_verts, int device) {
    return create_scene(nodes, n_nodes, n_nodes, prims, n_prims, verts, n_verts, nullptr, 0, device, nullptr, normals,
                        prim_alpha, uvs);
}

nnbvh_scene *nnbvh_scene_create_with_normals(const nnbvh_linear_node *nodes, int n_nodes, const nnbvh_prim *prims,
                                             int n_prims, const float *verts, const float *normals, int n_verts,
                                             int device) {
    return create_scene(nodes, n_nodes, n_nodes, prims, n_prims, verts, n_verts, nullptr, 0, device, nullptr, normals);
}

nnbvh_scene *nnbvh_scene_create_instanced(const nnbvh_linear_node *nodes, int n_nodes,
                                          int n_top_nodes, const nnbvh_prim *prims, int n_prims,
                                          const float *verts, int n_verts,
                                          const nnbvh_instance *instances, int n_instances,
                                          int device) {
    return create_scene(nodes, n_nodes, n_top_nodes, prims, n_prims, verts, n_verts, instances,
                        n_instances, device);
}

nnbvh_scene *nnbvh_scene_create_instanced_animated(const nnbvh_linear_node *nodes, int n_nodes,
                                                   int n_top_nodes, const nnbvh_prim *prims, int n_prims,
                                                   const float *verts, int n_verts,
                                                   const nnbvh_instance *instances, int n_instances,
                                                   const nnbvh_animated_transform *animated, int device) {
    if (animated)
        for (int k = 0; k < n_instances; ++k)
            if (animated[k].actually_animated && !(animated[k].end_time > animated[k].start_time)) {
                set_error("scene_create: animated instance with an empty time range");
                return nullptr;
            }
    return create_scene(nodes, n_nodes, n_top_nodes, prims, n_prims, verts, n_verts, instances,
                        n_instances, device, animated);
}

nnbvh_scene *nnbvh_scene_create_instanced_with_attributes(const nnbvh_linear_node *nodes, int n_nodes, int n_top_nodes,
                                                          const nnbvh_prim *prims, int n_prims, const float *verts,
                                                          int n_verts, const nnbvh_instance *instances, int n_instances,
                                                          const nnbvh_animated_transform *animated, const float *normals,
                                                          const float *uvs, const float *prim_alpha, int device) {
    if (animated)
        for (int k = 0; k < n_instances; ++k)
            if (animated[k].actually_animated && !(animated[k].end_time > animated[k].start_time)) {
                set_error("scene_create: animated instance with an empty time range");
                return nullptr;
            }
    return create_scene(nodes, n_nodes, n_top_nodes, prims, n_prims, verts, n_verts, instances, n_instances, device,
                        animated, normals, prim_alpha, uvs);
}

void nnbvh_transform_bounds(const float m[12], const float in[6], float out[6]) {
    // Transform::operator()(const Bounds3f&), util/transform.cpp:134-139: union of the 8
    // transformed corners (Bounds3::Corner, vecmath.h:1284-1289; point transform
    // util/transform.h:310-319 with w == 1)
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
        mn[k] = std::numeric_limits<float>::max();
        mx[k] = std::numeric_limits<float>::lowest();
    }
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {in[(c & 1) ? 3 : 0], in[(c & 2) ? 4 : 1], in[(c & 4) ? 5 : 2]};
        for (int k = 0; k < 3; ++k) {
            const float v = m[4 * k] * p[0] + m[4 * k + 1] * p[1] + m[4 * k + 2] * p[2] + m[4 * k + 3];
            mn[k] = std::min(mn[k], v);
            mx[k] = std::max(mx[k], v);
        }
    }
    std::memcpy(out, mn, 12);
    std::memcpy(out + 3, mx, 12);
}

void nnbvh_scene_destroy(nnbvh_scene *s) {
    if (!s) return;
    DeviceGuard guard(s->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : s->workspaces) {
        Workspace &w = kv.second;
        void *ptrs[] = {w.queue, w.spill, w.d_in, w.d_out, w.d_aux0, w.d_aux1};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        for (void *p : w.scratch)
            if (p) (void)hipFree(p);
    }
    for (HostSlot &sl : s->host_slots) {
        for (void *p : {sl.d_in, sl.d_out[0], sl.d_out[1], sl.d_out[2]})
            if (p) (void)hipFree(p);
        for (void *p : {sl.h_in, sl.h_out[0], sl.h_out[1], sl.h_out[2]})
            if (p) (void)hipHostFree(p);
        for (hipEvent_t e : {sl.ev_in, sl.ev_traced, sl.ev_out})
            if (e) (void)hipEventDestroy(e);
    }
    for (hipStream_t st : {s->host_up, s->host_trace, s->host_down})
        if (st) (void)hipStreamDestroy(st);
    for (int k = 0; k < nnbvh_scene::kSideStreams; ++k) {
        if (s->side[k]) (void)hipStreamDestroy(s->side[k]);
        if (s->ev_join[k]) (void)hipEventDestroy(s->ev_join[k]);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    (void)hipFree(s->d_wide);  // d_prims points into the same allocation
    if (s->d_anim) (void)hipFree(s->d_anim);
    if (s->d_stats) (void)hipFree(s->d_stats);
    delete s;
}

int nnbvh_scene_bounds(const nnbvh_scene *s, float out[6]) {
    if (!s || !out) {
        set_error("scene_bounds: null argument");
        return NNBVH_ERR_ARG;
    }
    std::memcpy(out, s->bounds, sizeof s->bounds);
    return NNBVH_OK;
}

// Persistent grid = what is resident at once (register/LDS limited), asked from the runtime
// for the exact kernel instance.  The kernel needs no co-residency (no grid barrier; late
// blocks just find less work in the queues), so a wrong answer costs speed, never results.
// both device arrays below 4 GiB (64 B per interior record, 16 B per primitive slot): the lean
// kernel instances reach them with 32-bit byte offsets
static int scene_fits32(const nnbvh_scene *s) {
    return (int64_t)s->n_interior < (1LL << 26) && s->n_slots < (1LL << 28) - 8;
}
// ... and the whole allocation (records, then the stream): a merged trip's one 32-bit offset reaches both
static int scene_prim_min(const nnbvh_scene *s) {
    const int64_t end = ((const char *)s->d_prims - (const char *)s->d_wide) + s->n_slots * 16 + 64;
    return end < (1LL << 32) ? s->prim_min : 0;
}

static int grid_blocks(nnbvh_scene *s, int mode) {
    int per_cu = s->blocks_per_cu;
    if (per_cu <= 0) {
        TraceParams dummy{};
        dummy.hasHostPrims = s->has_host_prims;  // selects between the lean and the general instances
        dummy.anim = s->d_anim;                  // ... and between the static- and the animated-instance ones
        dummy.fits32 = scene_fits32(s);
        int occ = 0;
        if (launch_trace(mode, dummy, s->window, s->instanced, patch_bits(s), 0, nullptr, &occ) != hipSuccess ||
            occ <= 0)
            occ = std::max(1, std::min(8, 160 / (s->window * 2)));
        per_cu = occ;
    }
    per_cu = std::min(per_cu, 8);
    return s->n_cus * per_cu;
}

int nnbvh_scene_info(const nnbvh_scene *s, int64_t out[6]) {
    if (!s || !out) {
        set_error("scene_info: null argument");
        return NNBVH_ERR_ARG;
    }
    out[0] = s->n_interior;
    out[1] = s->n_slots;
    out[2] = s->depth;
    out[3] = (int64_t)s->device_bytes;
    out[4] = grid_blocks(const_cast<nnbvh_scene *>(s), 0);
    out[5] = s->window;
    return NNBVH_OK;
}

int nnbvh_scene_sched_stats(nnbvh_scene *s, uint64_t out[16], int reset) {
    if (!s || !out) {
        set_error("scene_sched_stats: null argument");
        return NNBVH_ERR_ARG;
    }
    std::memset(out, 0, 16 * sizeof(uint64_t));
    if (!s->d_stats) return NNBVH_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    if (!hip_ok(hipDeviceSynchronize(), "scene_sched_stats") ||
        !hip_ok(hipMemcpy(out, s->d_stats, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost),
                "scene_sched_stats"))
        return NNBVH_ERR_DEVICE;
    if (reset) (void)hipMemset(s->d_stats, 0, 16 * sizeof(unsigned long long));
    return NNBVH_OK;
}

int nnbvh_scene_set_option(nnbvh_scene *s, const char *key, int value) {
    if (!s || !key) {
        set_error("set_option: null argument");
        return NNBVH_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(s->mu);
    std::string k(key);
    if (k == "stack_window") {
        if (value != 4 && value != 8 && value != 16) {
            set_error("set_option: stack_window must be 4, 8 or 16");
            return NNBVH_ERR_ARG;
        }
        s->window = value;
    } else if (k == "blocks_per_cu") {
        if (value < 0 || value > 8) {
            set_error("set_option: blocks_per_cu must be 0..8");
            return NNBVH_ERR_ARG;
        }
        s->blocks_per_cu = value;
    } else if (k == "int_repeat") {
        if (value < 1 || value > 16) {
            set_error("set_option: int_repeat must be 1..16");
            return NNBVH_ERR_ARG;
        }
        s->int_repeat = value;
    } else if (k == "host_chunk") {
        if (value < 1024) {
            set_error("set_option: host_chunk must be at least 1024 rays");
            return NNBVH_ERR_ARG;
        }
        s->host_chunk = value;
    } else if (k == "prim_min") {
        if (value < 0 || value > 64) {
            set_error("set_option: prim_min must be 0..64");
            return NNBVH_ERR_ARG;
        }
        s->prim_min = value;
    } else if (k == "prim_repeat") {
        if (value < 1 || value > 16) {
            set_error("set_option: prim_repeat must be 1..16");
            return NNBVH_ERR_ARG;
        }
        s->prim_repeat = value;
    } else if (k == "fused_batches") {
        s->fused_batches = value ? 1 : 0;
    } else if (k == "xcd_queues") {
        s->xcd_queues = value ? 1 : 0;
    } else if (k == "prim_weight") {
        if (value < 1 || value > 64) {
            set_error("set_option: prim_weight must be 1..64");
            return NNBVH_ERR_ARG;
        }
        s->prim_weight = value;
    } else if (k == "refill_weight") {
        if (value < 1 || value > 64) {
            set_error("set_option: refill_weight must be 1..64");
            return NNBVH_ERR_ARG;
        }
        s->refill_weight = value;
    } else {
        set_error("set_option: unknown key");
        return NNBVH_ERR_ARG;
    }
    return NNBVH_OK;
}

}  // extern "C"

// One workspace per stream: launches on one stream are ordered, so they may share the
// queue heads and the spill array; launches on different streams get their own.
static Workspace *workspace_for(nnbvh_scene *s, hipStream_t stream) {
    auto it = s->workspaces.find(stream);
    if (it != s->workspaces.end()) return &it->second;
    Workspace w;
    // an entry is spilled only when W - 1 newer ones sit above it (W >= 4): levels 0 .. depth - 3 of a lane's list
    const size_t spill_bytes = (size_t)std::max(s->depth - 2, 1) * (size_t)s->max_grid_threads * sizeof(uint2);
    if (!hip_ok(hipMalloc((void **)&w.queue, kMaxFusedBatches * kMaxQueues * kQueueStrideWords * sizeof(unsigned)),
                "hipMalloc(queue)"))
        return nullptr;
    if (!hip_ok(hipMalloc((void **)&w.spill, spill_bytes), "hipMalloc(spill)")) {
        (void)hipFree(w.queue);
        return nullptr;
    }
    return &(s->workspaces[stream] = w);
}

static int launch(nnbvh_scene *s, int mode, const void *d_rays, int64_t n, void *d_hits,
                  void *d_occ, void *d_vis, void *d_tests, hipStream_t stream, Workspace *w,
                  const int32_t *d_n = nullptr, const nnbvh_ray_soa *soa = nullptr) {
    TraceParams p{};
    if (soa) p.soa = *soa;  // d_rays == nullptr: the kernel reads the queue's SOA slices itself
    p.wide = s->d_wide;
    p.prims = s->d_prims;
    std::memcpy(p.rootMin, s->bounds, 12);
    std::memcpy(p.rootMax, s->bounds + 3, 12);
    p.rootRef = s->root_ref;
    p.rays = (const nnbvh_ray *)d_rays;
    p.hits = (nnbvh_hit *)d_hits;
    p.occluded = (uint8_t *)d_occ;
    p.visitedOut = (int32_t *)d_vis;
    p.testsOut = (int32_t *)d_tests;
    p.n = (long)n;
    p.nDev = d_n;
    p.queue = w->queue;
    p.nQueues = s->xcd_queues ? kMaxQueues : 1;
    p.primWeight = s->prim_weight;
    p.refillWeight = s->refill_weight;
    p.stats = s->d_stats;
    p.intRepeat = s->int_repeat;
    p.primRepeat = s->prim_repeat;
    p.fits32 = scene_fits32(s);
    p.primsOff = (unsigned)((const char *)s->d_prims - (const char *)s->d_wide);
    p.primMin = scene_prim_min(s);
    p.hasHostPrims = s->has_host_prims;
    p.spill = w->spill;
    p.anim = s->d_anim;
    p.nBatches = 0;
    p.anyMask = 0;
    if (!hip_ok(launch_zero_queue(w->queue, kMaxQueues * kQueueStrideWords, stream), "queue reset launch"))
        return NNBVH_ERR_DEVICE;
    // never launch more threads than there are rays to start with (tiny batches)
    int blocks = grid_blocks(s, mode);
    const int64_t need = (n + kBlockThreads - 1) / kBlockThreads;
    if (need < blocks) blocks = (int)std::max<int64_t>(need, 1);
    if (!hip_ok(launch_trace(mode, p, s->window, s->instanced, patch_bits(s), blocks, stream, nullptr),
                "trace kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

// the lean kernel instances (bvh_trace.hip) have forms that read a wavefront queue's SOA slices themselves
static bool scene_runs_lean(const nnbvh_scene *s) {
    return !s->instanced && patch_bits(s) == 0 && !s->has_host_prims && scene_fits32(s) && s->window == 8;
}

static bool grow(void **ptr, size_t *have, size_t need, const char *what) {
    if (*have >= need) return true;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    if (!hip_ok(hipMalloc(ptr, need), what)) return false;
    *have = need;
    return true;
}

extern "C" {

int nnbvh_intersect_closest_device(nnbvh_scene *s, const void *d_rays, int64_t n, void *d_hits,
                                   void *stream) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_hits))) {
        set_error("intersect_closest_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    if (n >= 0x7fffffffLL) {
        set_error("intersect_closest_device: at most 2^31-1 rays per call");
        return NNBVH_ERR_ARG;
    }
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    Workspace *w = workspace_for(s, (hipStream_t)stream);
    if (!w) return NNBVH_ERR_DEVICE;
    return launch(s, 0, d_rays, n, d_hits, nullptr, nullptr, nullptr, (hipStream_t)stream, w);
}

int nnbvh_intersect_any_device(nnbvh_scene *s, const void *d_rays, int64_t n, void *d_occluded,
                               void *d_nodes_visited, void *d_prim_tests, void *stream) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_occluded))) {
        set_error("intersect_any_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    if (n >= 0x7fffffffLL) {
        set_error("intersect_any_device: at most 2^31-1 rays per call");
        return NNBVH_ERR_ARG;
    }
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    Workspace *w = workspace_for(s, (hipStream_t)stream);
    if (!w) return NNBVH_ERR_DEVICE;
    const int mode = (d_nodes_visited || d_prim_tests) ? 1 : 2;
    return launch(s, mode, d_rays, n, nullptr, d_occluded, d_nodes_visited, d_prim_tests,
                  (hipStream_t)stream, w);
}

// One mode-3 launch over up to kMaxFusedBatches closest-hit / occlusion-only batches (the caller has checked that
// the scene and the batches allow it).  d_n: nullable array of nullable device-resident batch sizes.
static int launch_fused_batches(nnbvh_scene *s, Workspace *w, hipStream_t stream, const nnbvh_batch *batches,
                                int n_batches, const int32_t *const *d_n, const nnbvh_ray_soa *const *soas = nullptr) {
    TraceParams p{};
    p.wide = s->d_wide;
    p.prims = s->d_prims;
    std::memcpy(p.rootMin, s->bounds, 12);
    std::memcpy(p.rootMax, s->bounds + 3, 12);
    p.rootRef = s->root_ref;
    p.queue = w->queue;
    p.nQueues = s->xcd_queues ? kMaxQueues : 1;
    p.primWeight = s->prim_weight;
    p.refillWeight = s->refill_weight;
    p.stats = s->d_stats;
    p.intRepeat = s->int_repeat;
    p.primRepeat = s->prim_repeat;
    p.fits32 = scene_fits32(s);
    p.primsOff = (unsigned)((const char *)s->d_prims - (const char *)s->d_wide);
    p.primMin = scene_prim_min(s);
    p.hasHostPrims = s->has_host_prims;
    p.spill = w->spill;
    p.anim = s->d_anim;
    int64_t total = 0;
    for (int i = 0; i < n_batches; ++i) {
        if (batches[i].n == 0) continue;  // empty batches take no slot
        const int b = p.nBatches++;
        p.bRays[b] = (const nnbvh_ray *)batches[i].d_rays;
        p.bOut[b] = batches[i].d_out;
        p.bN[b] = (long)batches[i].n;
        p.bNDev[b] = d_n ? d_n[i] : nullptr;
        if (soas && soas[i]) p.bSoa[b] = *soas[i];  // with d_rays == nullptr: read as SOA slices
        if (batches[i].kind == NNBVH_BATCH_ANY) p.anyMask |= 1u << b;
        total += batches[i].n;
    }
    if (p.nBatches == 0) return NNBVH_OK;
    p.n = (long)total;
    if (!hip_ok(launch_zero_queue(w->queue, kMaxFusedBatches * kMaxQueues * kQueueStrideWords, stream),
                "queue reset launch"))
        return NNBVH_ERR_DEVICE;
    int blocks = grid_blocks(s, 3);
    const int64_t need = (total + kBlockThreads - 1) / kBlockThreads;
    if (need < blocks) blocks = (int)std::max<int64_t>(need, 1);
    if (!hip_ok(launch_trace(3, p, s->window, s->instanced, patch_bits(s), blocks, stream, nullptr),
                "fused trace kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

static bool batches_fusable(const nnbvh_scene *s, const nnbvh_batch *batches, int n_batches) {
    bool fusable = s->fused_batches && n_batches <= kMaxFusedBatches && s->window == 8 && !s->has_alpha;
    for (int i = 0; fusable && i < n_batches; ++i)
        fusable = batches[i].n < (1LL << kFusedIndexBits) &&
                  !(batches[i].kind == NNBVH_BATCH_ANY && (batches[i].d_nodes_visited || batches[i].d_prim_tests));
    return fusable;
}

int nnbvh_trace_batches_device(nnbvh_scene *s, const nnbvh_batch *batches, int n_batches,
                               void *stream_) {
    if (!s || n_batches < 0 || (n_batches > 0 && !batches)) {
        set_error("trace_batches_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    for (int i = 0; i < n_batches; ++i) {
        const nnbvh_batch &b = batches[i];
        if ((b.kind != NNBVH_BATCH_CLOSEST && b.kind != NNBVH_BATCH_ANY) || b.n < 0 ||
            b.n >= 0x7fffffffLL || (b.n > 0 && (!b.d_rays || !b.d_out))) {
            set_error("trace_batches_device: bad batch (kind, size or null buffer)");
            return NNBVH_ERR_ARG;
        }
    }
    if (n_batches == 0) return NNBVH_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = (hipStream_t)stream_;
    // One launch for all batches (mode 3) when they are closest-hit / occlusion-only any-hit batches of
    // fewer than 2^28 rays each: they share one ramp-up and one drain instead of paying one each.
    const bool fusable = batches_fusable(s, batches, n_batches);
    if (fusable) {
        Workspace *w = workspace_for(s, stream);
        if (!w) return NNBVH_ERR_DEVICE;
        return launch_fused_batches(s, w, stream, batches, n_batches, nullptr);
    }
    if (!s->ev_fork) {
        bool ok = hip_ok(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming), "hipEventCreate");
        for (int k = 0; ok && k < nnbvh_scene::kSideStreams; ++k)
            ok = hip_ok(hipStreamCreateWithFlags(&s->side[k], hipStreamNonBlocking), "hipStreamCreate") &&
                 hip_ok(hipEventCreateWithFlags(&s->ev_join[k], hipEventDisableTiming), "hipEventCreate");
        if (!ok) return NNBVH_ERR_DEVICE;
    }
    if (!hip_ok(hipEventRecord(s->ev_fork, stream), "hipEventRecord(fork)")) return NNBVH_ERR_DEVICE;
    bool used[nnbvh_scene::kSideStreams] = {false, false, false, false};
    for (int i = 0; i < n_batches; ++i) {
        const nnbvh_batch &b = batches[i];
        if (b.n == 0) continue;
        const int k = i % nnbvh_scene::kSideStreams;
        if (!used[k]) {
            if (!hip_ok(hipStreamWaitEvent(s->side[k], s->ev_fork, 0), "hipStreamWaitEvent(fork)"))
                return NNBVH_ERR_DEVICE;
            used[k] = true;
        }
        Workspace *w = workspace_for(s, s->side[k]);
        if (!w) return NNBVH_ERR_DEVICE;
        int rc;
        if (b.kind == NNBVH_BATCH_CLOSEST)
            rc = launch(s, 0, b.d_rays, b.n, b.d_out, nullptr, nullptr, nullptr, s->side[k], w);
        else
            rc = launch(s, (b.d_nodes_visited || b.d_prim_tests) ? 1 : 2, b.d_rays, b.n, nullptr,
                        b.d_out, b.d_nodes_visited, b.d_prim_tests, s->side[k], w);
        if (rc != NNBVH_OK) return rc;
    }
    for (int k = 0; k < nnbvh_scene::kSideStreams; ++k) {
        if (!used[k]) continue;
        if (!hip_ok(hipEventRecord(s->ev_join[k], s->side[k]), "hipEventRecord(join)") ||
            !hip_ok(hipStreamWaitEvent(stream, s->ev_join[k], 0), "hipStreamWaitEvent(join)"))
            return NNBVH_ERR_DEVICE;
    }
    return NNBVH_OK;
}

// ---- wavefront queues (wavefront/aggregate.cpp:34-68 on the device) ---------------------------
static bool soa_ok(const nnbvh_ray_soa *q) {
    return q && q->ox && q->oy && q->oz && q->dx && q->dy && q->dz;
}

int nnbvh_wavefront_intersect_closest(nnbvh_scene *s, int32_t max_rays, const nnbvh_ray_soa *ray_queue,
                                      const int32_t *d_size, const uint8_t *d_prim_class,
                                      int64_t n_prim_class, void *d_hits,
                                      const nnbvh_closest_queues *out, void *stream_) {
    if (!s || max_rays < 0 || !out || (max_rays > 0 && (!soa_ok(ray_queue) || !d_hits)) ||
        n_prim_class < 0) {
        set_error("wavefront_intersect_closest: bad argument");
        return NNBVH_ERR_ARG;
    }
    const nnbvh_work_queue *qs[6] = {&out->escaped, &out->hit_area_light, &out->basic_eval_material,
                                     &out->universal_eval_material, &out->medium_sample,
                                     &out->next_ray};
    for (const nnbvh_work_queue *q : qs)
        if (q->size && (q->capacity < 0 || (q->capacity > 0 && !q->items))) {
            set_error("wavefront_intersect_closest: queue with a size counter but no item storage");
            return NNBVH_ERR_ARG;
        }
    if (max_rays == 0) return NNBVH_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = (hipStream_t)stream_;
    Workspace *w = workspace_for(s, stream);
    if (!w) return NNBVH_ERR_DEVICE;
    const WavefrontCount cnt{max_rays, d_size};
    const int max_blocks = s->n_cus * 8;
    int rc;
    if (scene_runs_lean(s)) {
        // the traversal kernel reads the queue's SOA slices itself (no gather pass into nnbvh_ray records)
        rc = launch(s, 0, nullptr, max_rays, d_hits, nullptr, nullptr, nullptr, stream, w, d_size, ray_queue);
    } else {
        if (!grow(&w->d_in, &w->in_bytes, (size_t)max_rays * sizeof(nnbvh_ray), "hipMalloc(wavefront rays)"))
            return NNBVH_ERR_DEVICE;
        if (!hip_ok(launch_wf_gather(*ray_queue, cnt, w->d_in, max_blocks, stream), "gather kernel launch"))
            return NNBVH_ERR_DEVICE;
        rc = launch(s, 0, w->d_in, max_rays, d_hits, nullptr, nullptr, nullptr, stream, w, d_size);
    }
    if (rc != NNBVH_OK) return rc;
    if (!hip_ok(launch_wf_enqueue_closest(d_hits, cnt, ray_queue->has_medium, d_prim_class,
                                          (long)n_prim_class, *out, max_blocks, stream),
                "enqueue kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_wavefront_intersect_shadow(nnbvh_scene *s, int32_t max_rays, const nnbvh_ray_soa *shadow_queue,
                                     const int32_t *d_size, const float *d_Ld, const float *d_r_u,
                                     const float *d_r_l, const int32_t *d_pixel_index, float *d_L,
                                     int64_t n_pixels, uint8_t *d_occluded, void *stream_) {
    if (!s || max_rays < 0 || n_pixels < 0 ||
        (max_rays > 0 && (!soa_ok(shadow_queue) || !d_Ld || !d_r_u || !d_r_l || !d_pixel_index || !d_L))) {
        set_error("wavefront_intersect_shadow: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (max_rays == 0) return NNBVH_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = (hipStream_t)stream_;
    Workspace *w = workspace_for(s, stream);
    if (!w) return NNBVH_ERR_DEVICE;
    uint8_t *occ = d_occluded;
    if (!occ) {
        if (!grow(&w->d_out, &w->out_bytes, (size_t)max_rays, "hipMalloc(wavefront occluded)"))
            return NNBVH_ERR_DEVICE;
        occ = (uint8_t *)w->d_out;
    }
    const WavefrontCount cnt{max_rays, d_size};
    const int max_blocks = s->n_cus * 8;
    int rc;
    if (scene_runs_lean(s)) {
        rc = launch(s, 2, nullptr, max_rays, nullptr, occ, nullptr, nullptr, stream, w, d_size, shadow_queue);
    } else {
        if (!grow(&w->d_in, &w->in_bytes, (size_t)max_rays * sizeof(nnbvh_ray), "hipMalloc(wavefront rays)"))
            return NNBVH_ERR_DEVICE;
        if (!hip_ok(launch_wf_gather(*shadow_queue, cnt, w->d_in, max_blocks, stream), "gather kernel launch"))
            return NNBVH_ERR_DEVICE;
        rc = launch(s, 2, w->d_in, max_rays, nullptr, occ, nullptr, nullptr, stream, w, d_size);
    }
    if (rc != NNBVH_OK) return rc;
    if (!hip_ok(launch_wf_record_shadow(occ, cnt, d_Ld, d_r_u, d_r_l, d_pixel_index, d_L, (long)n_pixels,
                                        max_blocks, stream),
                "shadow record kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

// IntersectShadow of one depth and IntersectClosest of the next (wavefront/integrator.cpp: TraceShadowRays(depth),
// then the next iteration's IntersectClosest): both queues are filled by the shading of the same depth and neither
// reads what the other writes, so they can share ONE launch (mode 3: one ramp-up and one drain instead of two).
int nnbvh_wavefront_intersect_closest_and_shadow(
    nnbvh_scene *s, int32_t max_rays, const nnbvh_ray_soa *ray_queue, const int32_t *d_size,
    const uint8_t *d_prim_class, int64_t n_prim_class, void *d_hits, const nnbvh_closest_queues *out,
    int32_t max_shadow_rays, const nnbvh_ray_soa *shadow_queue, const int32_t *d_shadow_size, const float *d_Ld,
    const float *d_r_u, const float *d_r_l, const int32_t *d_pixel_index, float *d_L, int64_t n_pixels,
    uint8_t *d_occluded, void *stream_) {
    // a scene or sizes the one-launch form does not cover, or one side empty: the two calls one after the other
    nnbvh_batch probe[2] = {{NNBVH_BATCH_CLOSEST, 0, nullptr, max_rays, nullptr, nullptr, nullptr},
                            {NNBVH_BATCH_ANY, 0, nullptr, max_shadow_rays, nullptr, nullptr, nullptr}};
    if (!s || max_rays <= 0 || max_shadow_rays <= 0 || !batches_fusable(s, probe, 2)) {
        int rc = nnbvh_wavefront_intersect_shadow(s, max_shadow_rays, shadow_queue, d_shadow_size, d_Ld, d_r_u, d_r_l,
                                                  d_pixel_index, d_L, n_pixels, d_occluded, stream_);
        if (rc != NNBVH_OK) return rc;
        return nnbvh_wavefront_intersect_closest(s, max_rays, ray_queue, d_size, d_prim_class, n_prim_class, d_hits,
                                                 out, stream_);
    }
    if (!out || !soa_ok(ray_queue) || !d_hits || n_prim_class < 0 || n_pixels < 0 || !soa_ok(shadow_queue) || !d_Ld ||
        !d_r_u || !d_r_l || !d_pixel_index || !d_L) {
        set_error("wavefront_intersect_closest_and_shadow: bad argument");
        return NNBVH_ERR_ARG;
    }
    const nnbvh_work_queue *qs[6] = {&out->escaped, &out->hit_area_light, &out->basic_eval_material,
                                     &out->universal_eval_material, &out->medium_sample, &out->next_ray};
    for (const nnbvh_work_queue *q : qs)
        if (q->size && (q->capacity < 0 || (q->capacity > 0 && !q->items))) {
            set_error("wavefront_intersect_closest_and_shadow: queue with a size counter but no item storage");
            return NNBVH_ERR_ARG;
        }
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = (hipStream_t)stream_;
    Workspace *w = workspace_for(s, stream);
    if (!w) return NNBVH_ERR_DEVICE;
    uint8_t *occ = d_occluded;
    if (!occ) {
        if (!grow(&w->d_out, &w->out_bytes, (size_t)max_shadow_rays, "hipMalloc(wavefront occluded)"))
            return NNBVH_ERR_DEVICE;
        occ = (uint8_t *)w->d_out;
    }
    const WavefrontCount cnt{max_rays, d_size}, scnt{max_shadow_rays, d_shadow_size};
    const int max_blocks = s->n_cus * 8;
    // the (longer) closest-hit batch first: the shadow rays fill the lanes its tail leaves idle
    const int32_t *sizes[2] = {d_size, d_shadow_size};
    int rc;
    if (scene_runs_lean(s)) {  // both queues are read as the SOA slices they are
        const nnbvh_batch batches[2] = {{NNBVH_BATCH_CLOSEST, 0, nullptr, max_rays, d_hits, nullptr, nullptr},
                                        {NNBVH_BATCH_ANY, 0, nullptr, max_shadow_rays, occ, nullptr, nullptr}};
        const nnbvh_ray_soa *soas[2] = {ray_queue, shadow_queue};
        rc = launch_fused_batches(s, w, stream, batches, 2, sizes, soas);
    } else {
        const size_t closest_bytes = (size_t)max_rays * sizeof(nnbvh_ray);
        if (!grow(&w->d_in, &w->in_bytes, closest_bytes + (size_t)max_shadow_rays * sizeof(nnbvh_ray),
                  "hipMalloc(wavefront rays)"))
            return NNBVH_ERR_DEVICE;
        void *closest_rays = w->d_in, *shadow_rays = (char *)w->d_in + closest_bytes;
        if (!hip_ok(launch_wf_gather(*ray_queue, cnt, closest_rays, max_blocks, stream), "gather kernel launch") ||
            !hip_ok(launch_wf_gather(*shadow_queue, scnt, shadow_rays, max_blocks, stream), "gather kernel launch"))
            return NNBVH_ERR_DEVICE;
        const nnbvh_batch batches[2] = {{NNBVH_BATCH_CLOSEST, 0, closest_rays, max_rays, d_hits, nullptr, nullptr},
                                        {NNBVH_BATCH_ANY, 0, shadow_rays, max_shadow_rays, occ, nullptr, nullptr}};
        rc = launch_fused_batches(s, w, stream, batches, 2, sizes);
    }
    if (rc != NNBVH_OK) return rc;
    if (!hip_ok(launch_wf_enqueue_closest(d_hits, cnt, ray_queue->has_medium, d_prim_class, (long)n_prim_class, *out,
                                          max_blocks, stream),
                "enqueue kernel launch") ||
        !hip_ok(launch_wf_record_shadow(occ, scnt, d_Ld, d_r_u, d_r_l, d_pixel_index, d_L, (long)n_pixels, max_blocks,
                                        stream),
                "shadow record kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_wavefront_record_shadow_device(const uint8_t *d_occluded, int32_t max_rays, const int32_t *d_size,
                                         const float *d_Ld, const float *d_r_u, const float *d_r_l,
                                         const int32_t *d_pixel_index, float *d_L, int64_t n_pixels,
                                         int device, void *stream_) {
    if (max_rays < 0 || n_pixels < 0 ||
        (max_rays > 0 && (!d_occluded || !d_Ld || !d_r_u || !d_r_l || !d_pixel_index || !d_L))) {
        set_error("wavefront_record_shadow_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (max_rays == 0) return NNBVH_OK;
    DeviceGuard guard(device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    const WavefrontCount cnt{max_rays, d_size};
    if (!hip_ok(launch_wf_record_shadow(d_occluded, cnt, d_Ld, d_r_u, d_r_l, d_pixel_index, d_L,
                                        (long)n_pixels, 256 * 8, (hipStream_t)stream_),
                "shadow record kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

// ---- Triangle::InteractionFromIntersection post-pass (shapes.h:884-1010) --------------------------
static_assert(sizeof(nnbvh_interaction) == 192, "nnbvh_interaction must be 192 bytes");

}  // extern "C"

struct nnbvh_shading_mesh {
    int device = 0;
    int n_cus = 0;
    ShadingMeshDevice d;
};

template <typename T>
static bool upload(T **dst, const T *src, size_t count, const char *what) {
    *dst = nullptr;
    if (!src || count == 0) return true;
    return hip_ok(hipMalloc((void **)dst, count * sizeof(T)), what) &&
           hip_ok(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice), what);
}

extern "C" {

nnbvh_shading_mesh *nnbvh_shading_mesh_create(const float *verts, int n_verts,
                                              const int32_t *tri_vertices,
                                              const int32_t *patch_vertices, int n_tris,
                                              const float *normals, const float *uvs,
                                              const float *tangents, const int32_t *face_indices,
                                              const uint8_t *tri_flags, int device) {
    if (!verts || !tri_vertices || n_verts <= 0 || n_tris <= 0) {
        set_error("shading_mesh_create: empty vertex or triangle array");
        return nullptr;
    }
    for (long i = 0; i < 3L * n_tris; ++i) {
        const int v = tri_vertices[i];
        const bool not_a_triangle = tri_vertices[i - i % 3] < 0;
        if (!not_a_triangle && (v < 0 || v >= n_verts)) {
            set_error("shading_mesh_create: vertex index out of range");
            return nullptr;
        }
    }
    for (long i = 0; patch_vertices && i < 4L * n_tris; ++i) {
        const int v = patch_vertices[i];
        const bool not_a_patch = patch_vertices[i - i % 4] < 0;
        if (!not_a_patch && (v < 0 || v >= n_verts)) {
            set_error("shading_mesh_create: patch vertex index out of range");
            return nullptr;
        }
    }
    DeviceGuard guard(device);
    if (!guard.ok) return nullptr;
    auto *m = new nnbvh_shading_mesh;
    m->device = device;
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) {
        delete m;
        return nullptr;
    }
    m->n_cus = prop.multiProcessorCount;
    m->d.nTris = n_tris;
    m->d.nVerts = n_verts;
    m->d.defaultFlags = (uvs ? NNBVH_TRI_HAS_UV : 0) | (normals ? NNBVH_TRI_HAS_N : 0) |
                        (tangents ? NNBVH_TRI_HAS_S : 0);
    const bool ok = upload(&m->d.verts, verts, 3 * (size_t)n_verts, "shading mesh: vertices") &&
                    upload(&m->d.triVerts, tri_vertices, 3 * (size_t)n_tris, "shading mesh: indices") &&
                    upload(&m->d.patchVerts, patch_vertices, 4 * (size_t)n_tris, "shading mesh: patch indices") &&
                    upload(&m->d.normals, normals, 3 * (size_t)n_verts, "shading mesh: normals") &&
                    upload(&m->d.uvs, uvs, 2 * (size_t)n_verts, "shading mesh: uvs") &&
                    upload(&m->d.tangents, tangents, 3 * (size_t)n_verts, "shading mesh: tangents") &&
                    upload(&m->d.faceIndices, face_indices, (size_t)n_tris, "shading mesh: face indices") &&
                    upload(&m->d.triFlags, tri_flags, (size_t)n_tris, "shading mesh: flags");
    if (!ok) {
        nnbvh_shading_mesh_destroy(m);
        return nullptr;
    }
    return m;
}

int nnbvh_shading_mesh_set_instances(nnbvh_shading_mesh *m, const nnbvh_instance *instances, int n_instances) {
    if (!m || n_instances < 0 || (n_instances > 0 && !instances)) {
        set_error("shading_mesh_set_instances: bad argument");
        return NNBVH_ERR_ARG;
    }
    DeviceGuard guard(m->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    if (m->d.instances) (void)hipFree(m->d.instances);
    m->d.instances = nullptr;
    m->d.nInstances = 0;
    if (n_instances == 0) return NNBVH_OK;
    if (!upload(&m->d.instances, instances, (size_t)n_instances, "shading mesh: instances")) return NNBVH_ERR_DEVICE;
    m->d.nInstances = n_instances;
    return NNBVH_OK;
}

int nnbvh_shading_mesh_set_instances_animated(nnbvh_shading_mesh *m, const nnbvh_instance *instances,
                                              const nnbvh_animated_transform *animated, int n_instances) {
    int rc = nnbvh_shading_mesh_set_instances(m, instances, n_instances);
    if (rc != NNBVH_OK) return rc;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    for (float **p : {&m->d.anim, &m->d.animFwd}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (!animated || n_instances == 0) return NNBVH_OK;
    std::vector<float> table((size_t)n_instances * kAnimStride, 0.0f), fwd((size_t)n_instances * 24);
    for (int k = 0; k < n_instances; ++k) {
        fill_anim_entry(animated[k], &table[(size_t)k * kAnimStride]);
        std::memcpy(&fwd[(size_t)k * 24], animated[k].start_from, 48);       // rows 0..2 of startTransform.m
        std::memcpy(&fwd[(size_t)k * 24 + 12], animated[k].end_from, 48);    // ... of endTransform.m
    }
    if (!upload(&m->d.anim, table.data(), table.size(), "shading mesh: animation table") ||
        !upload(&m->d.animFwd, fwd.data(), fwd.size(), "shading mesh: animation table"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

void nnbvh_shading_mesh_destroy(nnbvh_shading_mesh *m) {
    if (!m) return;
    DeviceGuard guard(m->device);
    if (m->d.instances) (void)hipFree(m->d.instances);
    if (m->d.anim) (void)hipFree(m->d.anim);
    if (m->d.animFwd) (void)hipFree(m->d.animFwd);
    void *ptrs[] = {m->d.verts, m->d.triVerts, m->d.patchVerts, m->d.normals, m->d.uvs, m->d.tangents, m->d.faceIndices, m->d.triFlags};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
}

int nnbvh_triangle_interactions_device(const nnbvh_shading_mesh *m, const void *d_rays,
                                       const nnbvh_ray_soa *ray_soa, const void *d_hits,
                                       int32_t max_items, const int32_t *d_size, void *d_out,
                                       void *stream) {
    const bool soa_given = ray_soa && ray_soa->dx && ray_soa->dy && ray_soa->dz;
    if (!m || max_items < 0 || (max_items > 0 && (!d_hits || !d_out || (!d_rays && !soa_given)))) {
        set_error("triangle_interactions_device: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (max_items == 0) return NNBVH_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    if (!hip_ok(launch_triangle_interactions(m->d, d_rays, d_rays ? nullptr : ray_soa, d_hits, max_items, d_size,
                                             d_out, m->n_cus * 8, (hipStream_t)stream),
                "interaction kernel launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_triangle_interactions(const nnbvh_shading_mesh *m, const nnbvh_ray *rays, const nnbvh_hit *hits,
                                int32_t n, nnbvh_interaction *out) {
    if (!m || n < 0 || (n > 0 && (!rays || !hits || !out))) {
        set_error("triangle_interactions: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    DeviceGuard guard(m->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    void *d_rays = nullptr, *d_hits = nullptr, *d_out = nullptr;
    int rc = NNBVH_ERR_DEVICE;
    if (hip_ok(hipMalloc(&d_rays, (size_t)n * sizeof(nnbvh_ray)), "hipMalloc(rays)") &&
        hip_ok(hipMalloc(&d_hits, (size_t)n * sizeof(nnbvh_hit)), "hipMalloc(hits)") &&
        hip_ok(hipMalloc(&d_out, (size_t)n * sizeof(nnbvh_interaction)), "hipMalloc(interactions)") &&
        hip_ok(hipMemcpy(d_rays, rays, (size_t)n * sizeof(nnbvh_ray), hipMemcpyHostToDevice), "copy rays") &&
        hip_ok(hipMemcpy(d_hits, hits, (size_t)n * sizeof(nnbvh_hit), hipMemcpyHostToDevice), "copy hits") &&
        hip_ok(launch_triangle_interactions(m->d, d_rays, nullptr, d_hits, n, nullptr, d_out, m->n_cus * 8, nullptr),
               "interaction kernel launch") &&
        hip_ok(hipMemcpy(out, d_out, (size_t)n * sizeof(nnbvh_interaction), hipMemcpyDeviceToHost), "copy interactions"))
        rc = NNBVH_OK;
    for (void *p : {d_rays, d_hits, d_out})
        if (p) (void)hipFree(p);
    return rc;
}

// ---- IntersectShadowTr / IntersectOneRandom (wavefront/aggregate.cpp:70-116), media-free -------------
// Host-driven loops of device passes; the only host round trip per pass is the 4-byte count of
// items that go on (interface surfaces are rare: the usual shadow batch ends after its first pass).
static bool scratch(Workspace *w, int slot, size_t bytes, void **out) {
    if (!grow(&w->scratch[slot], &w->scratch_bytes[slot], std::max<size_t>(bytes, 16), "hipMalloc(wavefront scratch)"))
        return false;
    *out = w->scratch[slot];
    return true;
}

static bool read_count(const int32_t *d_counter, hipStream_t stream, int *out) {
    int32_t v = 0;
    if (!hip_ok(hipMemcpyAsync(&v, d_counter, 4, hipMemcpyDeviceToHost, stream), "read pass count") ||
        !hip_ok(hipStreamSynchronize(stream), "wavefront pass"))
        return false;
    *out = v;
    return true;
}

int nnbvh_wavefront_intersect_shadow_tr(nnbvh_scene *s, const nnbvh_shading_mesh *m, int32_t max_rays,
                                        const nnbvh_ray_soa *shadow_queue, const int32_t *d_size,
                                        const uint8_t *d_prim_class, int64_t n_prim_class, const float *d_Ld,
                                        const float *d_r_u, const float *d_r_l, const int32_t *d_pixel_index,
                                        float *d_L, int64_t n_pixels, uint8_t *d_state, void *stream_) {
    if (!s || !m || max_rays < 0 || n_pixels < 0 || n_prim_class < 0 ||
        (max_rays > 0 && (!soa_ok(shadow_queue) || !d_Ld || !d_r_u || !d_r_l || !d_pixel_index || !d_L))) {
        set_error("wavefront_intersect_shadow_tr: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (m->device != s->device) {
        set_error("wavefront_intersect_shadow_tr: scene and shading mesh live on different devices");
        return NNBVH_ERR_ARG;
    }
    if (max_rays == 0) return NNBVH_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = (hipStream_t)stream_;
    Workspace *w = workspace_for(s, stream);
    if (!w) return NNBVH_ERR_DEVICE;
    const size_t n = (size_t)max_rays;
    void *raysA, *raysB, *hitsA, *hitsB, *origA, *origB, *pLight, *state, *counters, *intr;
    if (!scratch(w, 0, n * 32, &raysA) || !scratch(w, 1, n * 32, &raysB) || !scratch(w, 2, n * 32, &hitsA) ||
        !scratch(w, 3, n * 32, &hitsB) || !scratch(w, 4, n * 4, &origA) || !scratch(w, 5, n * 4, &origB) ||
        !scratch(w, 6, n * 16, &pLight) || !scratch(w, 7, n, &state) || !scratch(w, 8, 64, &counters))
        return NNBVH_ERR_DEVICE;
    int32_t *nCur = (int32_t *)counters, *nNext = nCur + 1;
    const WavefrontCount cnt{max_rays, d_size};
    const int max_blocks = s->n_cus * 8;
    if (!hip_ok(launch_str_init(*shadow_queue, cnt, raysA, (int32_t *)origA, (float4 *)pLight, (uint8_t *)state,
                                max_blocks, stream), "shadow-tr init launch"))
        return NNBVH_ERR_DEVICE;
    // the first pass covers the whole queue: its size is max_rays clamped by *d_size
    if (d_size) {
        if (!hip_ok(hipMemcpyAsync(nCur, d_size, 4, hipMemcpyDeviceToDevice, stream), "copy queue size"))
            return NNBVH_ERR_DEVICE;
    } else if (!hip_ok(hipMemcpyAsync(nCur, &max_rays, 4, hipMemcpyHostToDevice, stream), "copy queue size")) {
        return NNBVH_ERR_DEVICE;
    }
    int active = max_rays;
    for (int pass = 0; active > 0; ++pass) {
        if (pass > 4096) {
            set_error("wavefront_intersect_shadow_tr: more than 4096 interface surfaces on one shadow ray");
            return NNBVH_ERR_ARG;
        }
        int rc = launch(s, 0, raysA, active, hitsA, nullptr, nullptr, nullptr, stream, w, nCur);
        if (rc != NNBVH_OK) return rc;
        if (!hip_ok(hipMemsetAsync(nNext, 0, 4, stream), "reset pass count") ||
            !hip_ok(launch_str_classify(raysA, hitsA, (const int32_t *)origA, nCur, d_prim_class, (long)n_prim_class,
                                        (uint8_t *)state, raysB, hitsB, (int32_t *)origB, nNext, active, max_blocks,
                                        stream), "shadow-tr classify launch"))
            return NNBVH_ERR_DEVICE;
        int n_iface = 0;
        if (!read_count(nNext, stream, &n_iface)) return NNBVH_ERR_DEVICE;
        if (n_iface <= 0) break;
        if (!scratch(w, 9, (size_t)n_iface * sizeof(nnbvh_interaction), &intr)) return NNBVH_ERR_DEVICE;
        if (!hip_ok(launch_triangle_interactions(m->d, raysB, nullptr, hitsB, n_iface, nNext, intr, m->n_cus * 8, stream),
                    "interaction kernel launch") ||
            !hip_ok(hipMemsetAsync(nCur, 0, 4, stream), "reset pass count") ||
            !hip_ok(launch_str_spawn(raysB, intr, (const int32_t *)origB, nNext, (const float4 *)pLight, (uint8_t *)state,
                                     raysA, (int32_t *)origA, nCur, n_iface, max_blocks, stream), "shadow-tr spawn launch"))
            return NNBVH_ERR_DEVICE;
        if (!read_count(nCur, stream, &active)) return NNBVH_ERR_DEVICE;
    }
    if (!hip_ok(launch_str_record((const uint8_t *)state, cnt, d_Ld, d_r_u, d_r_l, d_pixel_index, d_L, (long)n_pixels,
                                  d_state, max_blocks, stream), "shadow-tr record launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_wavefront_intersect_one_random(nnbvh_scene *s, const nnbvh_shading_mesh *m, int32_t max_items,
                                         const float *d_p0, const float *d_p1, const int32_t *d_material,
                                         const int32_t *d_size, const int32_t *d_prim_material,
                                         int64_t n_prim_material, void *d_sel_hits, void *d_sel_rays,
                                         float *d_reservoir_pdf, float *d_weight_sum, void *stream_) {
    if (!s || !m || max_items < 0 || n_prim_material < 0 ||
        (max_items > 0 && (!d_p0 || !d_p1 || !d_material || !d_sel_hits || !d_sel_rays || !d_reservoir_pdf))) {
        set_error("wavefront_intersect_one_random: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (m->device != s->device) {
        set_error("wavefront_intersect_one_random: scene and shading mesh live on different devices");
        return NNBVH_ERR_ARG;
    }
    if (max_items == 0) return NNBVH_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    hipStream_t stream = (hipStream_t)stream_;
    Workspace *w = workspace_for(s, stream);
    if (!w) return NNBVH_ERR_DEVICE;
    const size_t n = (size_t)max_items;
    void *raysA, *raysB, *hits, *origA, *origB, *pi, *rng, *weights, *counters, *intr;
    if (!scratch(w, 0, n * 32, &raysA) || !scratch(w, 1, n * 32, &raysB) || !scratch(w, 2, n * 32, &hits) ||
        !scratch(w, 4, n * 4, &origA) || !scratch(w, 5, n * 4, &origB) || !scratch(w, 6, n * 36, &pi) ||
        !scratch(w, 10, n * 16, &rng) || !scratch(w, 11, n * 8, &weights) || !scratch(w, 8, 64, &counters))
        return NNBVH_ERR_DEVICE;
    int32_t *nCur = (int32_t *)counters, *nNext = nCur + 1;
    OneRandomState st{(float *)pi, (uint64_t *)rng, (float *)weights};
    const WavefrontCount cnt{max_items, d_size};
    const int max_blocks = s->n_cus * 8;
    if (!hip_ok(hipMemsetAsync(nCur, 0, 4, stream), "reset pass count") ||
        !hip_ok(launch_or_init(d_p0, d_p1, cnt, st, raysA, (int32_t *)origA, nCur, d_sel_hits, d_sel_rays, max_blocks,
                               stream), "one-random init launch"))
        return NNBVH_ERR_DEVICE;
    int active = 0;
    if (!read_count(nCur, stream, &active)) return NNBVH_ERR_DEVICE;
    void *cur = raysA, *next = raysB, *ocur = origA, *onext = origB;
    for (int pass = 0; active > 0; ++pass) {
        if (pass > 65536) {
            set_error("wavefront_intersect_one_random: more than 65536 surfaces on one segment");
            return NNBVH_ERR_ARG;
        }
        int rc = launch(s, 0, cur, active, hits, nullptr, nullptr, nullptr, stream, w, nCur);
        if (rc != NNBVH_OK) return rc;
        if (!scratch(w, 9, (size_t)active * sizeof(nnbvh_interaction), &intr)) return NNBVH_ERR_DEVICE;
        if (!hip_ok(launch_triangle_interactions(m->d, cur, nullptr, hits, active, nCur, intr, m->n_cus * 8, stream),
                    "interaction kernel launch") ||
            !hip_ok(hipMemsetAsync(nNext, 0, 4, stream), "reset pass count") ||
            !hip_ok(launch_or_step(cur, hits, intr, (const int32_t *)ocur, nCur, d_p1, d_material, d_prim_material,
                                   (long)n_prim_material, st, next, (int32_t *)onext, nNext, d_sel_hits, d_sel_rays,
                                   active, max_blocks, stream), "one-random step launch"))
            return NNBVH_ERR_DEVICE;
        if (!read_count(nNext, stream, &active)) return NNBVH_ERR_DEVICE;
        std::swap(cur, next);
        std::swap(ocur, onext);
        std::swap(nCur, nNext);
    }
    if (!hip_ok(launch_or_finish(cnt, st, d_reservoir_pdf, d_weight_sum, max_blocks, stream), "one-random finish launch"))
        return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

// ---- host-buffer entry points: a pipeline of chunks -------------------------------------------------
// What Integrator::Intersect / IntersectP callers (cpu/integrators.cpp:296-313) hand over lives in host
// memory.  The batch is cut into chunks of up to kHostChunk rays that rotate over kHostSlots slots, each
// with its own stream, device buffers and traversal workspace: chunk k's rays go up while chunk k-1 is
// traced and chunk k-2's results come down.  Memory the caller has pinned (hipHostMalloc / hipHostRegister,
// nnbvh_host_register) is read and written by the copy engines directly; pageable memory goes through pinned
// staging buffers filled / drained by a few host threads while the other slots' GPU work is in flight.
// Every ray's result is what the single-shot path gives (rays are independent).
static bool host_is_pinned(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary (unregistered) host pointer: not an error
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

static void parallel_copy(void *dst, const void *src, size_t bytes) {
    constexpr size_t kPiece = 4u << 20;
    const int pieces = (int)std::min<size_t>(4, bytes / kPiece);
    if (pieces < 2) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t each = (bytes / (size_t)pieces + 63) & ~(size_t)63;
    std::thread helpers[3];
    for (int k = 1; k < pieces; ++k) {
        const size_t off = each * (size_t)k, len = (k == pieces - 1) ? bytes - off : each;
        helpers[k - 1] = std::thread([=] { std::memcpy((char *)dst + off, (const char *)src + off, len); });
    }
    std::memcpy(dst, src, each);
    for (int k = 1; k < pieces; ++k) helpers[k - 1].join();
}

struct HostArray {  // one per-ray output array of a call
    void *host;
    size_t elem;  // bytes per ray
    bool pinned;
};

static bool slot_reserve(HostSlot &sl, size_t rays, bool need_staging_in, const HostArray *outs, int n_outs) {
    for (hipEvent_t *e : {&sl.ev_in, &sl.ev_traced, &sl.ev_out})
        if (!*e && !hip_ok(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate")) return false;
    auto dev = [&](void **p, size_t *have, size_t need) {
        if (*have >= need) return true;
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *have = 0;
        if (!hip_ok(hipMalloc(p, need), "hipMalloc(host pipeline)")) return false;
        *have = need;
        return true;
    };
    auto pin = [&](void **p, size_t *have, size_t need) {
        if (*have >= need) return true;
        if (*p) (void)hipHostFree(*p);
        *p = nullptr;
        *have = 0;
        if (!hip_ok(hipHostMalloc(p, need, hipHostMallocDefault), "hipHostMalloc(host pipeline)")) return false;
        *have = need;
        return true;
    };
    if (!dev(&sl.d_in, &sl.d_in_bytes, rays * 32)) return false;
    if (need_staging_in && !pin(&sl.h_in, &sl.h_in_bytes, rays * 32)) return false;
    for (int k = 0; k < n_outs; ++k) {
        if (!dev(&sl.d_out[k], &sl.d_out_bytes[k], rays * outs[k].elem)) return false;
        if (outs[k].host && !outs[k].pinned && !pin(&sl.h_out[k], &sl.h_out_bytes[k], rays * outs[k].elem)) return false;
    }
    return true;
}

// mode 0: outs = {hits}; mode 1 / 2: outs = {occluded, nodes_visited?, prim_tests?} (absent arrays: host = null)
//
// Three streams with fixed roles — upload, trace, download — and per-slot events between them.  The copies have
// their own streams on purpose: a copy queued on the stream of the kernel it depends on is performed by a copy
// KERNEL, which has to wait for compute units behind the next chunk's persistent trace kernel (measured: no
// overlap at all); a copy on a stream of its own goes to a DMA engine and overlaps the trace
// (tools/overlap_copy_probe.py: trace + upload = upload alone).
static int host_pipeline(nnbvh_scene *s, int mode, const nnbvh_ray *rays, int64_t n, HostArray *outs, int n_outs) {
    for (hipStream_t *st : {&s->host_up, &s->host_trace, &s->host_down})
        if (!*st && !hip_ok(hipStreamCreateWithFlags(st, hipStreamNonBlocking), "hipStreamCreate")) return NNBVH_ERR_DEVICE;
    const bool rays_pinned = host_is_pinned(rays);
    for (int k = 0; k < n_outs; ++k) outs[k].pinned = outs[k].host && host_is_pinned(outs[k].host);
    // chunks: a launch costs ~0.5 ms of ramp-up and drain whatever its size (DESIGN.md "why launches are large"), so
    // a batch is cut into at most kHostChunks chunks of at least host_chunk rays
    int64_t chunk = std::max<int64_t>(s->host_chunk, (n + nnbvh_scene::kHostChunks - 1) / nnbvh_scene::kHostChunks);
    chunk = std::min<int64_t>(chunk, n);
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    Workspace *w = workspace_for(s, s->host_trace);
    if (!w) return NNBVH_ERR_DEVICE;
    struct Pending {
        int64_t first = 0, count = 0;
        bool busy = false;
    } pending[nnbvh_scene::kHostSlots];
    auto drain = [&](int slot) -> bool {  // wait for the slot's chunk and hand its staged results to the caller
        Pending &pd = pending[slot];
        if (!pd.busy) return true;
        HostSlot &sl = s->host_slots[slot];
        if (!hip_ok(hipEventSynchronize(sl.ev_out), "host pipeline")) return false;
        for (int k = 0; k < n_outs; ++k)
            if (outs[k].host && !outs[k].pinned)
                parallel_copy((char *)outs[k].host + (size_t)pd.first * outs[k].elem, sl.h_out[k], (size_t)pd.count * outs[k].elem);
        pd.busy = false;
        return true;
    };
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int slot = (int)(c % nnbvh_scene::kHostSlots);
        if (!drain(slot)) return NNBVH_ERR_DEVICE;
        HostSlot &sl = s->host_slots[slot];
        const int64_t first = c * chunk, count = std::min<int64_t>(chunk, n - first);
        if (!slot_reserve(sl, (size_t)chunk, !rays_pinned, outs, n_outs)) return NNBVH_ERR_DEVICE;
        const void *src = rays + first;
        if (!rays_pinned) {
            parallel_copy(sl.h_in, rays + first, (size_t)count * 32);
            src = sl.h_in;
        }
        if (!hip_ok(hipMemcpyAsync(sl.d_in, src, (size_t)count * 32, hipMemcpyHostToDevice, s->host_up), "copy rays") ||
            !hip_ok(hipEventRecord(sl.ev_in, s->host_up), "host pipeline") ||
            !hip_ok(hipStreamWaitEvent(s->host_trace, sl.ev_in, 0), "host pipeline"))
            return NNBVH_ERR_DEVICE;
        const int rc = mode == 0 ? launch(s, 0, sl.d_in, count, sl.d_out[0], nullptr, nullptr, nullptr, s->host_trace, w)
                                 : launch(s, mode, sl.d_in, count, nullptr, sl.d_out[0], n_outs > 1 ? sl.d_out[1] : nullptr,
                                          n_outs > 2 ? sl.d_out[2] : nullptr, s->host_trace, w);
        if (rc != NNBVH_OK) return rc;
        if (!hip_ok(hipEventRecord(sl.ev_traced, s->host_trace), "host pipeline") ||
            !hip_ok(hipStreamWaitEvent(s->host_down, sl.ev_traced, 0), "host pipeline"))
            return NNBVH_ERR_DEVICE;
        for (int k = 0; k < n_outs; ++k) {
            if (!outs[k].host) continue;
            void *dst = outs[k].pinned ? (void *)((char *)outs[k].host + (size_t)first * outs[k].elem) : sl.h_out[k];
            if (!hip_ok(hipMemcpyAsync(dst, sl.d_out[k], (size_t)count * outs[k].elem, hipMemcpyDeviceToHost, s->host_down),
                        "copy results"))
                return NNBVH_ERR_DEVICE;
        }
        if (!hip_ok(hipEventRecord(sl.ev_out, s->host_down), "host pipeline")) return NNBVH_ERR_DEVICE;
        pending[slot].first = first;
        pending[slot].count = count;
        pending[slot].busy = true;
    }
    for (int64_t c = n_chunks; c < n_chunks + nnbvh_scene::kHostSlots; ++c)  // oldest first
        if (!drain((int)(c % nnbvh_scene::kHostSlots))) return NNBVH_ERR_DEVICE;
    return NNBVH_OK;
}

int nnbvh_intersect_closest(nnbvh_scene *s, const nnbvh_ray *rays, int64_t n, nnbvh_hit *hits) {
    if (!s || n < 0 || (n > 0 && (!rays || !hits))) {
        set_error("intersect_closest: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    if (n >= 0x7fffffffLL) {
        set_error("intersect_closest: at most 2^31-1 rays per call");
        return NNBVH_ERR_ARG;
    }
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);  // host path: one call at a time per scene
    HostArray outs[1] = {{hits, 32, false}};
    return host_pipeline(s, 0, rays, n, outs, 1);
}

int nnbvh_intersect_any(nnbvh_scene *s, const nnbvh_ray *rays, int64_t n, uint8_t *occluded,
                        int32_t *nodes_visited, int32_t *prim_tests) {
    if (!s || n < 0 || (n > 0 && (!rays || !occluded))) {
        set_error("intersect_any: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (n == 0) return NNBVH_OK;
    if (n >= 0x7fffffffLL) {
        set_error("intersect_any: at most 2^31-1 rays per call");
        return NNBVH_ERR_ARG;
    }
    DeviceGuard guard(s->device);
    if (!guard.ok) return NNBVH_ERR_DEVICE;
    std::lock_guard<std::mutex> lock(s->mu);
    const bool counts = nodes_visited || prim_tests;
    // with counts the kernel writes both arrays; one the caller did not ask for stays on the device
    HostArray outs[3] = {{occluded, 1, false}, {nodes_visited, 4, false}, {prim_tests, 4, false}};
    return host_pipeline(s, counts ? 1 : 2, rays, n, outs, counts ? 3 : 1);
}

int nnbvh_host_register(void *ptr, size_t bytes) {
    if (!ptr || bytes == 0) {
        set_error("host_register: bad argument");
        return NNBVH_ERR_ARG;
    }
    if (reinterpret_cast<uintptr_t>(ptr) % 4096 != 0) {
        // a registration covers whole pages: a buffer that shares its first page with other heap objects would
        // leave THEIR memory mapped into the GPU's address space (include/nnbvh.h)
        set_error("host_register: the buffer must be page-aligned (4096) and own its pages");
        return NNBVH_ERR_ARG;
    }
    return hip_ok(hipHostRegister(ptr, bytes, hipHostRegisterDefault), "hipHostRegister") ? NNBVH_OK : NNBVH_ERR_DEVICE;
}

int nnbvh_host_unregister(void *ptr) {
    if (!ptr) {
        set_error("host_unregister: bad argument");
        return NNBVH_ERR_ARG;
    }
    return hip_ok(hipHostUnregister(ptr), "hipHostUnregister") ? NNBVH_OK : NNBVH_ERR_DEVICE;
}

}  // extern "C"
