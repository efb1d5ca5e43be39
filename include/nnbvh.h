/*
 * nnbvh.h — C ABI of the MI355X-native BVH traversal path (libnnbvh_hip.so).
 *
 * Drop-in boundary for pbrt's BVH hot path.  Each entry point names the reference
 * interface it replaces (paths relative to /root/reference/src/pbrt):
 *
 *   nnbvh_scene_create      <- CreateAccelerator("bvh", prims, ...) / new BVHAggregate(prims)
 *                              (cpu/aggregates.cpp:1163-1178, 140-190): takes the flattened
 *                              LinearBVHNode[] (cpu/aggregates.cpp:129-137), the leaf-ordered
 *                              primitive table (aggregates.cpp:176) and the mesh vertices
 *                              (util/mesh.h:24-72, already in render space, util/mesh.cpp:36-39)
 *   nnbvh_scene_bounds      <- BVHAggregate::Bounds()          (cpu/aggregates.cpp:524-527)
 *   nnbvh_intersect_closest <- BVHAggregate::Intersect()       (cpu/aggregates.cpp:529-579), batched
 *                              like CPUAggregate::IntersectClosest (wavefront/aggregate.cpp:34-58)
 *   nnbvh_intersect_any     <- BVHAggregate::IntersectP()      (cpu/aggregates.cpp:581-624), batched
 *                              like CPUAggregate::IntersectShadow  (wavefront/aggregate.cpp:60-68)
 *   nnbvh_wavefront_*       <- WavefrontAggregate::IntersectClosest / IntersectShadow incl. the
 *                              queue push rules (wavefront/integrator.h:32-54, intersect.h:16-156)
 *   nnbvh_triangle_interactions_device <- Triangle::InteractionFromIntersection (shapes.h:884-1010)
 *   nnbvh_build_*           <- BVHAggregate ctor + buildRecursive + flattenBVH
 *                              (cpu/aggregates.cpp:140-387, 505-522); host-side, no GPU needed
 *
 * Conventions: plain C, caller-owned buffers, no C++/torch types.  Every function that
 * can fail returns an int status (0 = NNBVH_OK) or NULL and records a message readable
 * with nnbvh_last_error() (thread-local); nothing here aborts the host process (the
 * reference's CHECK/LOG_FATAL would, util/check.h:36-57).  All entry points are
 * thread-safe; results of a call depend only on that call's inputs.  There is NO CPU
 * fallback: without a usable HIP device the intersect calls fail with NNBVH_ERR_DEVICE.
 */
#ifndef NNBVH_H
#define NNBVH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNBVH_OK 0
#define NNBVH_ERR_ARG 1     /* bad pointer / size / malformed tree */
#define NNBVH_ERR_DEVICE 2  /* no HIP device, allocation or launch failure */
#define NNBVH_ERR_DEPTH 3   /* tree deeper than the supported traversal stack */

/* == LinearBVHNode (cpu/aggregates.cpp:129-137): 32 B, 32-B aligned in the reference */
typedef struct nnbvh_linear_node {
    float pmin[3];
    float pmax[3];
    int32_t offset;   /* leaf: primitivesOffset; interior: secondChildOffset */
    uint16_t nprims;  /* 0 -> interior node */
    uint8_t axis;     /* interior: split axis 0/1/2 */
    uint8_t pad;
} nnbvh_linear_node;

#define NNBVH_PRIM_TRIANGLE 0       /* Triangle      (shapes.h:833-1192) */
#define NNBVH_PRIM_BILINEAR_PATCH 1 /* BilinearPatch (shapes.h:1350-1539): v = p00,p10,p01,p11 */
#define NNBVH_PRIM_HOST 3           /* a primitive only the host can intersect (Sphere, Disk, Cylinder,
                                       Curve, alpha-tested GeometricPrimitive: shapes.h, cpu/
                                       primitive.cpp:57-84): carries bounds only.  A ray that
                                       reaches one gets hit.instance = -1 (closest) or
                                       occluded = 2 unless a GPU primitive already occludes it
                                       (any): the caller re-traces exactly those rays on the CPU */
#define NNBVH_PRIM_INSTANCE 2       /* TransformedPrimitive (cpu/primitive.h:83-101): v[0] = index
                                       into the instance table; top-level tree only */

#define NNBVH_PRIM_ALPHA_TRIANGLE 4  /* a Triangle inside a GeometricPrimitive whose alpha texture is a
                                       constant (cpu/primitive.cpp:57-70, FloatConstantTexture): v[3] =
                                       the bit pattern of the float alpha.  A hit is ignored when
                                       HashFloat(r.o, r.d) > alpha (always when alpha <= 0); the reference
                                       then re-traces from the hit point against the same shape, which a
                                       planar triangle cannot be hit by again (shapes_test.cpp:156-206) —
                                       the device performs that re-test too and, should it ever hit, voids
                                       the ray like a host-only primitive (instance = -1 / occluded = 2).
                                       For meshes WITHOUT per-vertex shading normals (the offset normal
                                       is the geometric one); alpha-tested meshes with normals and
                                       textured alpha stay NNBVH_PRIM_HOST */
#define NNBVH_PRIM_ALPHA_TRIANGLE_FLIPPED 5 /* same, mesh->reverseOrientation ^ transformSwapsHandedness */
#define NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH 6  /* ... of a mesh WITH per-vertex shading normals: the ray re-traced
                                       after a rejected hit is offset along FaceForward(n, ns) (shapes.h:
                                       939-951, interaction.h:194-200); needs the scene's vertex normals
                                       (nnbvh_scene_create_with_normals; kd-trees:
                                       nnbvh_kd_scene_create_with_attributes) */
#define NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH_FLIPPED 7
#define NNBVH_PRIM_ALPHA_PATCH 8   /* BilinearPatch behind a GeometricPrimitive with a CONSTANT alpha (v[0..3] are the
                                       patch's vertices: the alpha comes from the per-primitive array of
                                       nnbvh_scene_create_with_attributes).  A non-planar patch can be met again by the
                                       ray re-traced off its own surface; the recursion of cpu/primitive.cpp:63-69 is
                                       followed for up to three re-traces, beyond which the record is void
                                       (needs-host).  For patch meshes WITHOUT (u, v) coordinates (with them: kinds
                                       12 .. 15) */
#define NNBVH_PRIM_ALPHA_PATCH_FLIPPED 9         /* same, mesh->reverseOrientation ^ transformSwapsHandedness */
#define NNBVH_PRIM_ALPHA_PATCH_SMOOTH 10         /* ... of a mesh WITH per-vertex normals (BilinearPatchMesh::n as
                                                     the mesh stores them, util/mesh.cpp:216-223) */
#define NNBVH_PRIM_ALPHA_PATCH_SMOOTH_FLIPPED 11
#define NNBVH_PRIM_ALPHA_PATCH_UV 12               /* 8 .. 11 for a mesh WITH (u, v) coordinates (BilinearPatchMesh::uv): the
                                                      interaction's normal is then Normalize(Cross(dpds, dpdt)) of the
                                                      (s, t) reparametrisation, shapes.h:1414-1437; needs the scene's
                                                      per-vertex uvs.  kind = 8 + flipped + 2 * smooth + 4 * uv */
#define NNBVH_PRIM_ALPHA_PATCH_UV_FLIPPED 13
#define NNBVH_PRIM_ALPHA_PATCH_UV_SMOOTH 14
#define NNBVH_PRIM_ALPHA_PATCH_UV_SMOOTH_FLIPPED 15

/* One entry of BVHAggregate::primitives: the shape handle flattened to global vertex
 * indices (Triangle{meshIndex,triIndex} -> mesh->vertexIndices[3*tri..], shapes.cpp:326-328). */
typedef struct nnbvh_prim {
    int32_t kind;
    int32_t id;   /* caller's primitive id, returned in nnbvh_hit.prim */
    int32_t v[4]; /* indices into the vertex array; v[3] unused for triangles */
} nnbvh_prim;

/* Ray + tMax (ray.h:18-39 Ray{o,d,time,medium} and the tMax argument), 32 B.
 * d need NOT be normalised (shadow rays are not, cpu/integrators.h:52-54). */
typedef struct nnbvh_ray {
    float o[3];
    float tmax;
    float d[3];
    float time;
} nnbvh_ray;

/* What TriangleIntersection / BilinearIntersection carry (shapes.h:820-824, 1271-1275)
 * plus the reference's observables bvhNodesVisited (aggregates.cpp:27) and nTriTests
 * (shapes.cpp:148), per ray.  32 B. */
typedef struct nnbvh_hit {
    int32_t prim; /* -1 = miss */
    float t;      /* tHit; on a miss: the ray's tmax */
    float b0, b1, b2; /* triangle barycentrics; patch: b0 = u, b1 = v, b2 = 0 */
    int32_t nodes_visited;
    int32_t prim_tests;
    int32_t instance; /* -1: the ray reached a host-only primitive: record void, re-trace on the CPU;
                         0: hit in the top-level tree (or miss); k + 1: inside instance k, and
                         then `prim` is the child tree's primitive id and t, b* are those of the
                         instance-space ray, exactly what TransformedPrimitive::Intersect returns */
} nnbvh_hit;

/* A TransformedPrimitive: a child BVHAggregate (an object instance, scene.cpp:1521-1577) behind
 * a static transform.  Matrices are the first three rows of pbrt's 4x4 Transform::m / mInv
 * (row-major; the fourth row of an affine transform is 0 0 0 1).  The child tree occupies
 * nodes[root, root + n_nodes) of the scene's node array, in the same DFS layout; its leaves
 * index the scene's primitive table.  Several instances may share one child tree. */
typedef struct nnbvh_instance {
    float render_from_prim[12];
    float prim_from_render[12];
    int32_t root;
    int32_t n_nodes;
} nnbvh_instance;

typedef struct nnbvh_scene nnbvh_scene;
typedef struct nnbvh_build nnbvh_build;

const char *nnbvh_last_error(void);
int nnbvh_device_count(void);

/* ---- host-side tree construction (no GPU) ---------------------------------------- */
#define NNBVH_SPLIT_SAH 0
#define NNBVH_SPLIT_HLBVH 1
#define NNBVH_SPLIT_MIDDLE 2
#define NNBVH_SPLIT_EQUAL_COUNTS 3
nnbvh_build *nnbvh_build_create(const nnbvh_prim *prims, int n_prims, const float *verts,
                                int n_verts, int max_prims_in_node, int split_method);
/* same, for a primitive list that contains NNBVH_PRIM_INSTANCE / NNBVH_PRIM_HOST entries:
 * prim_bounds holds 6 floats (min, max) per primitive and is read for those entries */
nnbvh_build *nnbvh_build_create_with_bounds(const nnbvh_prim *prims, int n_prims,
                                            const float *verts, int n_verts,
                                            const float *prim_bounds, int max_prims_in_node,
                                            int split_method);
/* Tree construction on the GPU (`device`), split_method NNBVH_SPLIT_SAH or NNBVH_SPLIT_HLBVH; the
 * result (nodes, leaf-ordered primitives, depth) is byte-identical to
 * nnbvh_build_create_with_bounds(..., split_method), i.e. to the reference's tree.
 *   SAH   (buildRecursive, aggregates.cpp:192-387): nodes above 256 primitives breadth-first with
 *         whole-grid kernels, every smaller subtree by one wavefront; std::partition's exact
 *         element order is reproduced from ballots / prefix sums.
 *   HLBVH (buildHLBVH, aggregates.cpp:389-503 with treelets emitted in Morton order): Morton codes,
 *         radix sort, the treelets' radix trees, bounds and the DFS layout on the device, only
 *         buildUpperSAH over the <= 4096 treelet roots (aggregates.cpp:626-723) on the host.
 * NULL + nnbvh_last_error() on failure (no device, allocation, malformed input). */
nnbvh_build *nnbvh_build_create_gpu(const nnbvh_prim *prims, int n_prims, const float *verts,
                                    int n_verts, const float *prim_bounds, int max_prims_in_node,
                                    int split_method, int device);
/* milliseconds of the last GPU build's phases (all zero for host builds) — HLBVH: upload, device
 * sort + tree, host upper SAH, device emit, download; SAH: upload, big nodes breadth-first,
 * wavefront subtrees, layout + bounds, download */
int nnbvh_build_gpu_timing(const nnbvh_build *b, double out_ms[5]);
const nnbvh_linear_node *nnbvh_build_nodes(const nnbvh_build *b, int *n_nodes);
const nnbvh_prim *nnbvh_build_ordered_prims(const nnbvh_build *b, int *n_prims);
int nnbvh_build_depth(const nnbvh_build *b); /* edges root -> deepest leaf */
void nnbvh_build_destroy(nnbvh_build *b);

/* ---- device scene ------------------------------------------------------------------ */
/* Uploads and bakes a flattened tree (the reference's LinearBVHNode[] in its DFS layout, first child
 * at index + 1) with its leaf-ordered primitives.  Everything the kernels index with is validated
 * here: a malformed tree is NULL + nnbvh_last_error(), never a device fault.  Node bounds must be
 * Bounds3f with pmin <= pmax on every axis and no NaN (what every builder emits; the kernels' form of
 * the slab test is equal to Bounds3::IntersectP, util/vecmath.h:1573-1608, for such boxes). */
nnbvh_scene *nnbvh_scene_create(const nnbvh_linear_node *nodes, int n_nodes,
                                const nnbvh_prim *ordered_prims, int n_prims,
                                const float *verts, int n_verts, int device);
/* two-level scene: nodes[0, n_top_nodes) is the top-level tree (may contain NNBVH_PRIM_INSTANCE
 * leaves), the rest are child trees named by `instances`.  Replaces the TransformedPrimitive
 * path of cpu/primitive.cpp:112-131 (AnimatedPrimitive is not covered). */
nnbvh_scene *nnbvh_scene_create_instanced(const nnbvh_linear_node *nodes, int n_nodes,
                                          int n_top_nodes, const nnbvh_prim *ordered_prims,
                                          int n_prims, const float *verts, int n_verts,
                                          const nnbvh_instance *instances, int n_instances,
                                          int device);
/* AnimatedPrimitive (cpu/primitive.h:103-118; cpu/primitive.cpp:133-158): an instance whose
 * render-from-instance transform is an AnimatedTransform.  The struct carries the members the
 * reference object holds after construction (util/transform.h:443-520: start / end transform, the
 * decomposition T, R, S of util/transform.cpp:375-394 and the time range); per ray the device evaluates
 * AnimatedTransform::Interpolate(ray.time) (util/transform.cpp:1062-1081) and applies its inverse as
 * for a static instance.  Arithmetic is the reference's operation for operation except the two sines of
 * Slerp, which the device evaluates in fp64 and rounds (libm's sinf differs from the correctly rounded
 * value on about one input in 10^5): the traversal path's one documented tolerance exception.
 * Bounds of such a primitive (AnimatedTransform::MotionBounds) come from the caller via prim_bounds,
 * like every instance's.  Hits inside animated instances get NNBVH_INTERACTION_HOST from the
 * interaction post-pass. */
typedef struct nnbvh_animated_transform {
    float start_from[16], start_inv[16]; /* startTransform m / mInv, row-major 4x4 */
    float end_from[16], end_inv[16];
    float T[2][3];
    float R[2][4];                       /* quaternion v.x v.y v.z w */
    float S[2][16];
    float start_time, end_time;
    int32_t actually_animated;           /* 0: a TransformedPrimitive (the nnbvh_instance matrices are used) */
    int32_t pad;
} nnbvh_animated_transform;
/* as nnbvh_scene_create_instanced; animated[k] (nullable array) belongs to instances[k] */
nnbvh_scene *nnbvh_scene_create_instanced_animated(const nnbvh_linear_node *nodes, int n_nodes,
                                                   int n_top_nodes, const nnbvh_prim *ordered_prims,
                                                   int n_prims, const float *verts, int n_verts,
                                                   const nnbvh_instance *instances, int n_instances,
                                                   const nnbvh_animated_transform *animated, int device);
/* ... and with the attributes the alpha-tested kinds read (nnbvh_scene_create_with_attributes): normals and uvs per
 * vertex, prim_alpha per entry of `prims`; animated and each attribute array may be NULL */
nnbvh_scene *nnbvh_scene_create_instanced_with_attributes(const nnbvh_linear_node *nodes, int n_nodes, int n_top_nodes,
                                                          const nnbvh_prim *prims, int n_prims, const float *verts,
                                                          int n_verts, const nnbvh_instance *instances, int n_instances,
                                                          const nnbvh_animated_transform *animated, const float *normals,
                                                          const float *uvs, const float *prim_alpha, int device);
/* Transform::operator()(const Bounds3f&) (util/transform.cpp:134-139): the bounds an instance
 * primitive presents to the top-level builder (TransformedPrimitive::Bounds, primitive.h:94) */
void nnbvh_transform_bounds(const float render_from_prim[12], const float in_min_max[6],
                            float out_min_max[6]);
/* Triangles in, traceable scene out, entirely on the device: the tree is built there
 * (nnbvh_build_create_gpu's builders), baked there into the traversal layout, and never visits the
 * host.  Same tree and same traversal results as nnbvh_build_create_with_bounds(...,
 * split_method) + nnbvh_scene_create.  split_method NNBVH_SPLIT_SAH or NNBVH_SPLIT_HLBVH;
 * triangles, bilinear patches and host-only primitives (instances: nnbvh_scene_create_instanced). */
nnbvh_scene *nnbvh_scene_create_gpu_build(const nnbvh_prim *prims, int n_prims, const float *verts,
                                          int n_verts, const float *prim_bounds,
                                          int max_prims_in_node, int split_method, int device);
/* ... with the attributes the alpha-tested kinds read (see nnbvh_scene_create_with_attributes below): normals and
 * uvs per vertex, prim_alpha per entry of `prims` (the CALLER's order: it is carried through the build); each may be NULL */
nnbvh_scene *nnbvh_scene_create_gpu_build_with_attributes(const nnbvh_prim *prims, int n_prims, const float *verts,
                                                          int n_verts, const float *prim_bounds, const float *normals,
                                                          const float *uvs, const float *prim_alpha,
                                                          int max_prims_in_node, int split_method, int device);
/* nnbvh_scene_create with the meshes' per-vertex shading normals (3 floats per vertex, indexed like `verts`;
 * TriangleMesh::n, util/mesh.h:48): read for NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH[_FLIPPED] primitives only, whose
 * three normals are baked into the primitive stream.  normals = NULL is nnbvh_scene_create. */
nnbvh_scene *nnbvh_scene_create_with_normals(const nnbvh_linear_node *nodes, int n_nodes, const nnbvh_prim *prims,
                                             int n_prims, const float *verts, const float *normals, int n_verts,
                                             int device);
/* ... and with the (u, v) coordinates per vertex (2 floats, read for the NNBVH_PRIM_ALPHA_PATCH_UV* primitives) and a
 * per-primitive constant alpha (n_prims floats, indexed like `prims`): read for the NNBVH_PRIM_ALPHA_PATCH*
 * primitives only (alpha-tested triangles keep theirs in v[3]).  Any of the three arrays may be NULL when no
 * primitive needs it. */
nnbvh_scene *nnbvh_scene_create_with_attributes(const nnbvh_linear_node *nodes, int n_nodes, const nnbvh_prim *prims,
                                                int n_prims, const float *verts, const float *normals,
                                                const float *uvs, const float *prim_alpha, int n_verts, int device);
void nnbvh_scene_destroy(nnbvh_scene *s);
int nnbvh_scene_bounds(const nnbvh_scene *s, float out_min_max[6]);
/* what the baked device layout looks like: [0]=interior records, [1]=prim-stream slots,
 * [2]=tree depth, [3]=device bytes, [4]=persistent grid blocks, [5]=LDS stack window */
int nnbvh_scene_info(const nnbvh_scene *s, int64_t out[6]);

/* ---- traversal, host buffers (synchronous; copies in and out) ----------------------
 * What Integrator::Intersect / IntersectP callers (cpu/integrators.cpp:296-313) hand over.  The batch is
 * traced in at most 6 chunks of at least "host_chunk" rays (option; default 2^20) that rotate over three
 * buffer slots, with an upload, a trace and a download stream: one chunk's rays go up while the previous one
 * is traced and the one before comes down.  Buffers
 * the caller has pinned (hipHostMalloc, hipHostRegister, nnbvh_host_register) are read and written by the
 * copy engines directly; pageable buffers go through pinned staging.  Results do not depend on the chunking. */
/* pin a caller buffer (hipHostRegister).  ptr must be PAGE-ALIGNED (4096) and the buffer should own its pages
 * (mmap, aligned_alloc / posix_memalign of whole pages): a registration covers whole pages, and pages shared with
 * other heap objects stay mapped into the GPU's address space on behalf of those neighbours — a later pageable
 * copy the runtime makes for THEM can then fault on the device (seen in round 3: "write access to a read-only
 * page" in an unrelated hipMemcpy after malloc'ed arrays had been registered and freed).  Unregister before the
 * memory is freed.  NNBVH_ERR_ARG for a misaligned pointer. */
int nnbvh_host_register(void *ptr, size_t bytes);
int nnbvh_host_unregister(void *ptr);
int nnbvh_intersect_closest(nnbvh_scene *s, const nnbvh_ray *rays, int64_t n, nnbvh_hit *hits);
/* nodes_visited / prim_tests may be NULL (then the faster non-counting kernel runs) */
int nnbvh_intersect_any(nnbvh_scene *s, const nnbvh_ray *rays, int64_t n, uint8_t *occluded,
                        int32_t *nodes_visited, int32_t *prim_tests);

/* ---- traversal, device buffers (asynchronous on `stream`, a hipStream_t; NULL = the
 *      default stream).  Pointers are device pointers on the scene's device. ----------- */
int nnbvh_intersect_closest_device(nnbvh_scene *s, const void *d_rays, int64_t n, void *d_hits,
                                   void *stream);
int nnbvh_intersect_any_device(nnbvh_scene *s, const void *d_rays, int64_t n, void *d_occluded,
                               void *d_nodes_visited, void *d_prim_tests, void *stream);

/* ---- several independent batches at once --------------------------------------------------
 * One call traces n_batches independent ray batches (e.g. the closest-hit queue and the
 * shadow-ray queue of one wavefront iteration: wavefront/integrator.cpp:403-406 and :575-579
 * are independent of each other).  Up to 4 batches of fewer than 2^28 rays each, closest-hit or
 * any-hit without counts, are traced by ONE kernel launch on `stream` (the persistent wavefronts
 * drain the batches one after the other, so the call pays one ramp-up and one drain — the
 * dependent chain of its longest ray — instead of one per batch).  Other
 * combinations run concurrently on internal streams forked from and joined back into `stream`.
 * Either way the call is ONE asynchronous operation on `stream`. */
#define NNBVH_BATCH_CLOSEST 0
#define NNBVH_BATCH_ANY 1
typedef struct nnbvh_batch {
    int32_t kind; /* NNBVH_BATCH_CLOSEST: d_out = nnbvh_hit[n]; NNBVH_BATCH_ANY: d_out = uint8[n] */
    int32_t pad;
    const void *d_rays;    /* nnbvh_ray[n], device */
    int64_t n;
    void *d_out;
    void *d_nodes_visited; /* ANY only, optional int32[n] (exact counts) */
    void *d_prim_tests;    /* ANY only, optional int32[n] */
} nnbvh_batch;
int nnbvh_trace_batches_device(nnbvh_scene *s, const nnbvh_batch *batches, int n_batches,
                               void *stream);

/* ---- wavefront queues: WavefrontAggregate::IntersectClosest / IntersectShadow ---------------
 * (wavefront/integrator.h:32-54; CPU implementation wavefront/aggregate.cpp:34-68).  The ray
 * queue arrives in the reference's SOA layout, its size may live on the device (the reference's
 * WorkQueue::size is read by the kernels, workqueue.h:54-63, 143-152), and the calls apply the
 * reference's enqueue rules (wavefront/intersect.h:16-30, 49-156) and shadow bookkeeping
 * (intersect.h:32-47) on the device.  Output queues hold INDICES into the input queue: the
 * caller's material / light / escape stages gather their payload (beta, lambda, pixelIndex ...)
 * from the input work items by index, and the geometric part of the hit from d_hits[index].
 * IntersectShadowTr and IntersectOneRandom (aggregate.cpp:70-116) follow below in their media-free
 * form; the media themselves (SampleT_maj along ray.medium) are outside this path's scope. */
typedef struct nnbvh_ray_soa {   /* SOA<Ray> slices (workitems.soa:40-50, 82-88), device pointers */
    const float *ox, *oy, *oz;
    const float *dx, *dy, *dz;
    const float *time;           /* nullable */
    const float *tmax;           /* ShadowRayWorkItem::tMax; NULL = Infinity (RayQueue) */
    const uint8_t *has_medium;   /* nullable: ray.medium != nullptr */
} nnbvh_ray_soa;

typedef struct nnbvh_work_queue { /* WorkQueue<T> (wavefront/workqueue.h:36-113) of item indices */
    int32_t *items;               /* device, `capacity` entries */
    int32_t *size;                /* device counter, advanced atomically; NULL = queue not wanted */
    int32_t capacity;             /* pushes beyond it are counted in *size but not stored */
    int32_t pad;
} nnbvh_work_queue;

typedef struct nnbvh_closest_queues { /* the out-parameters of IntersectClosest, in its order */
    nnbvh_work_queue escaped;                 /* EscapedRayQueue                               */
    nnbvh_work_queue hit_area_light;          /* HitAreaLightQueue                             */
    nnbvh_work_queue basic_eval_material;     /* MaterialEvalQueue, basic texture evaluator    */
    nnbvh_work_queue universal_eval_material; /* MaterialEvalQueue, universal evaluator        */
    nnbvh_work_queue medium_sample;           /* MediumSampleQueue                             */
    nnbvh_work_queue next_ray;                /* rays continuing through an interface surface  */
} nnbvh_closest_queues;

/* per-primitive class byte, indexed by the primitive id a hit returns: what the reference reads
 * off the hit's SurfaceInteraction when it enqueues (intersect.h:93-128) */
#define NNBVH_CLASS_BASIC 0      /* material evaluable by BasicTextureEvaluator                */
#define NNBVH_CLASS_UNIVERSAL 1  /* needs UniversalTextureEvaluator                           */
#define NNBVH_CLASS_INTERFACE 2  /* no material: interface between media (intersect.h:103)    */
#define NNBVH_CLASS_AREA_LIGHT 4 /* intr.areaLight set (intersect.h:113); combines with 0/1   */

/* n = min(max_rays, *d_size) when d_size != NULL, else max_rays.  d_hits: nnbvh_hit[max_rays].
 * d_prim_class may be NULL (every primitive NNBVH_CLASS_BASIC). */
int nnbvh_wavefront_intersect_closest(nnbvh_scene *s, int32_t max_rays, const nnbvh_ray_soa *ray_queue,
                                      const int32_t *d_size, const uint8_t *d_prim_class,
                                      int64_t n_prim_class, void *d_hits,
                                      const nnbvh_closest_queues *out, void *stream);
/* Unoccluded rays add Ld / (r_u + r_l).Average() to d_L[4 * pixel_index] (SampledSpectrum = 4
 * floats per item, SOA<SampledSpectrum> keeps them as one float4: util/soa.h:47-127).
 * d_occluded: optional uint8[max_rays] copy of the per-ray result. */
int nnbvh_wavefront_intersect_shadow(nnbvh_scene *s, int32_t max_rays, const nnbvh_ray_soa *shadow_queue,
                                     const int32_t *d_size, const float *d_Ld, const float *d_r_u,
                                     const float *d_r_l, const int32_t *d_pixel_index, float *d_L,
                                     int64_t n_pixels, uint8_t *d_occluded, void *stream);
/* IntersectShadow of depth d and IntersectClosest of depth d + 1 in ONE launch (wavefront/integrator.cpp: the
 * render loop traces the shadow rays of a depth and then the next depth's rays; both queues were filled by the same
 * shading pass and neither reads what the other writes).  Same results as the two calls; one ramp-up and one drain
 * of the traversal kernel instead of two.  Scenes the one-launch kernel does not cover (alpha-tested primitives)
 * and empty sides run the two calls one after the other. */
int nnbvh_wavefront_intersect_closest_and_shadow(
    nnbvh_scene *s, int32_t max_rays, const nnbvh_ray_soa *ray_queue, const int32_t *d_size,
    const uint8_t *d_prim_class, int64_t n_prim_class, void *d_hits, const nnbvh_closest_queues *out,
    int32_t max_shadow_rays, const nnbvh_ray_soa *shadow_queue, const int32_t *d_shadow_size, const float *d_Ld,
    const float *d_r_u, const float *d_r_l, const int32_t *d_pixel_index, float *d_L, int64_t n_pixels,
    uint8_t *d_occluded, void *stream);

/* WavefrontAggregate::IntersectShadowTr (wavefront/aggregate.cpp:70-88 -> TraceTransmittance,
 * wavefront/intersect.h:164-274) for scenes WITHOUT participating media: a shadow ray passes
 * through interface surfaces (NNBVH_CLASS_INTERFACE: no material) — closest hit, SpawnRayTo the light
 * point from the hit (ray.h:75-101), again — and is blocked by the first surface that has a material;
 * T_ray, r_u, r_l stay 1, so an arriving ray adds Ld * (1 / (r_u + r_l).Average()) to its pixel sample
 * (intersect.h:258-273).  Needs the scene's shading mesh for the hit points (pi, n).  Runs as passes
 * over the still-active rays; between passes the host reads one 4-byte count (not hipGraph-
 * capturable).  d_state: optional uint8[max_rays]: 0 = arrived, 1 = blocked, 2 = a host-only
 * primitive lies on the way (nothing added: the caller's to finish).  Media themselves (ray.medium,
 * SampleT_maj) are outside this path's scope. */
typedef struct nnbvh_shading_mesh nnbvh_shading_mesh;
int nnbvh_wavefront_intersect_shadow_tr(nnbvh_scene *s, const nnbvh_shading_mesh *m, int32_t max_rays,
                                        const nnbvh_ray_soa *shadow_queue, const int32_t *d_size,
                                        const uint8_t *d_prim_class, int64_t n_prim_class, const float *d_Ld,
                                        const float *d_r_u, const float *d_r_l, const int32_t *d_pixel_index,
                                        float *d_L, int64_t n_pixels, uint8_t *d_state, void *stream);
/* WavefrontAggregate::IntersectOneRandom (wavefront/aggregate.cpp:90-116; GPU form gpu/optix.cu:480-573):
 * for every SubsurfaceScatterWorkItem {p0, p1, material} walk the segment p0 -> p1 surface by surface
 * (SpawnRayTo(p1), Intersect(r, 1)) and keep ONE hit whose material equals the item's, chosen by
 * WeightedReservoirSampler (util/sampling.h:524-596, unit weights, PCG32 seeded with Hash(p0, p1)).
 * d_p0 / d_p1: 3 floats per item; d_material: the item's material id; d_prim_material: material id
 * per primitive id (NULL = all 0).  Out per item: the selected hit record (prim = -1: none; instance =
 * -1: a host-only primitive lies on the segment, the item is the caller's) and the ray of its segment
 * (what nnbvh_triangle_interactions_device needs to produce the SubsurfaceInteraction), the
 * reservoir's SampleProbability (0 without a sample) and, optionally, its weight sum. */
int nnbvh_wavefront_intersect_one_random(nnbvh_scene *s, const nnbvh_shading_mesh *m, int32_t max_items,
                                         const float *d_p0, const float *d_p1, const int32_t *d_material,
                                         const int32_t *d_size, const int32_t *d_prim_material,
                                         int64_t n_prim_material, void *d_sel_hits, void *d_sel_rays,
                                         float *d_reservoir_pdf, float *d_weight_sum, void *stream);

/* RecordShadowRayResult (wavefront/intersect.h:32-47) for a shadow batch that was traced with
 * nnbvh_intersect_any_device: the bookkeeping half of nnbvh_wavefront_intersect_shadow on its own. */
int nnbvh_wavefront_record_shadow_device(const uint8_t *d_occluded, int32_t max_rays, const int32_t *d_size,
                                         const float *d_Ld, const float *d_r_u, const float *d_r_l,
                                         const int32_t *d_pixel_index, float *d_L, int64_t n_pixels,
                                         int device, void *stream);

/* ---- film: RGBFilm's pixel accumulators on the device ------------------------------------------
 * RGBFilm::Pixel (film.h:302-307: double rgbSum[3], double weightSum; 32 B per pixel, row-major over
 * the pixel bounds) and RGBFilm::AddSample (film.h:239-255) as UpdateFilm calls it (wavefront/
 * film.cpp:13-40).  The spectral conversion sensor->ToSensorRGB(L, lambda) (film.h:95-100) stays with
 * the caller's sensor model: samples arrive as sensor RGB.  In a tile-sharded render every rank owns
 * the pixels of its tiles; pack / unpack move an index list of pixels between the film and a
 * contiguous buffer, which is what the RCCL all-gather of the per-tile film samples carries. */
typedef struct nnbvh_film nnbvh_film;
/* pixel bounds [x0, x1) x [y0, y1) (Film::PixelBounds); max_component_value = RGBFilm's clamp
 * ("maxcomponentvalue", default Infinity).  Pixels start at zero.  NULL + nnbvh_last_error(). */
nnbvh_film *nnbvh_film_create(int32_t x0, int32_t y0, int32_t x1, int32_t y1, float max_component_value,
                              int device);
void nnbvh_film_destroy(nnbvh_film *f);
int nnbvh_film_clear(nnbvh_film *f, void *stream);
/* Adds n_passes samples to each of n_per_pass pixel slots: slot i is pixel (d_px[i], d_py[i]) (slots
 * outside the bounds are skipped, wavefront/film.cpp:18-19; the slots' pixels must be distinct, as
 * the reference's pixelIndex is within a stage), sample (pass, i) is d_rgb[rgb_stride * (pass *
 * n_per_pass + i) + 0..2] with filter weight d_weight[pass * n_per_pass + i] (NULL = 1).  A slot's
 * passes are added in order, so the sums equal the reference's sample loop bit for bit.
 * n_per_pass is clamped to *d_size when d_size != NULL. */
int nnbvh_film_add_samples_device(nnbvh_film *f, const int32_t *d_px, const int32_t *d_py,
                                  const float *d_rgb, int32_t rgb_stride, const float *d_weight,
                                  int32_t n_per_pass, int32_t n_passes, const int32_t *d_size,
                                  void *stream);
/* the accumulators themselves: *d_pixels = double[4 * *n_pixels] on the film's device */
int nnbvh_film_pixels_device(nnbvh_film *f, void **d_pixels, int64_t *n_pixels);
int nnbvh_film_read(nnbvh_film *f, double *out); /* synchronous copy to double[4 * n_pixels] */
/* d_index: n linear pixel indices ((y - y0) * width + (x - x0)); buffer: 4 doubles per index */
int nnbvh_film_pack_pixels_device(nnbvh_film *f, const int32_t *d_index, int64_t n, void *d_out,
                                  void *stream);
int nnbvh_film_unpack_pixels_device(nnbvh_film *f, const int32_t *d_index, int64_t n, const void *d_in,
                                    void *stream);

/* ---- hit record -> SurfaceInteraction: Triangle:: / BilinearPatch::InteractionFromIntersection --
 * (shapes.h:884-1010 and 1396-1489, run by the shapes' Intersect on every reported hit; with
 * the SurfaceInteraction constructor and SetShadingGeometry, interaction.h:32-33, 164-214).  A
 * device post-pass over a batch of hit records; bit-identical to the reference function.
 * The mesh data are TriangleMesh's (util/mesh.h:24-72) flattened over all meshes of the scene, in
 * render space and as the TriangleMesh constructor stores them (util/mesh.cpp:36-64: normals
 * negated under reverseOrientation). */
#define NNBVH_TRI_FLIP_NORMAL 1 /* mesh->reverseOrientation ^ mesh->transformSwapsHandedness */
#define NNBVH_TRI_HAS_UV 2      /* the triangle's mesh has uv / n / s (meshes of one scene differ) */
#define NNBVH_TRI_HAS_N 4
#define NNBVH_TRI_HAS_S 8
/* tri_vertices: 3 vertex indices per primitive, primitive k being the one whose nnbvh_prim.id is k
 * (v[0] < 0: not a triangle).  patch_vertices (nullable): 4 per primitive, p00 p10 p01 p11
 * (v[0] < 0: not a bilinear patch; BilinearPatchMesh, util/mesh.h:50-72).  normals / tangents: 3
 * floats per vertex, uvs: 2, face_indices: 1 int per primitive; each nullable.  tri_flags:
 * NNBVH_TRI_* per primitive; NULL = no flip and the HAS_* bits follow from which arrays were given. */
nnbvh_shading_mesh *nnbvh_shading_mesh_create(const float *verts, int n_verts,
                                              const int32_t *tri_vertices,
                                              const int32_t *patch_vertices, int n_prims,
                                              const float *normals, const float *uvs,
                                              const float *tangents, const int32_t *face_indices,
                                              const uint8_t *tri_flags, int device);
/* optional: the instance table of a two-level scene (the one given to nnbvh_scene_create_instanced).
 * Hits inside instance k are then finished on the device as TransformedPrimitive::Intersect does
 * (cpu/primitive.cpp:112-125): interaction in the instance's space (wo = -ApplyInverse(ray.d)), then
 * Transform::operator()(const SurfaceInteraction &) with renderFromPrimitive (util/transform.cpp:
 * 229-261).  Without it such hits get NNBVH_INTERACTION_HOST. */
int nnbvh_shading_mesh_set_instances(nnbvh_shading_mesh *m, const nnbvh_instance *instances, int n_instances);
/* ... with AnimatedPrimitives among the instances (the tables given to nnbvh_scene_create_instanced_animated):
 * a hit inside an instance whose `animated[k].actually_animated` is set is finished as
 * AnimatedPrimitive::Intersect does (cpu/primitive.cpp:143-153) — the ray into the instance's space and the
 * interaction back out of it both through renderFromPrimitive.Interpolate(ray.time) (util/transform.cpp:
 * 1062-1081; the Slerp sines as in the traversal kernels).  animated = NULL: all static. */
int nnbvh_shading_mesh_set_instances_animated(nnbvh_shading_mesh *m, const nnbvh_instance *instances,
                                              const nnbvh_animated_transform *animated, int n_instances);
void nnbvh_shading_mesh_destroy(nnbvh_shading_mesh *m);

#define NNBVH_INTERACTION_MISS 0
#define NNBVH_INTERACTION_TRIANGLE 1 /* all fields valid (Triangle::InteractionFromIntersection) */
#define NNBVH_INTERACTION_HOST 2     /* hit on a host primitive, on a primitive the mesh has no vertices
                                        for, or inside an instance when no instance table was set: the
                                        caller finishes it on the host; only prim and status are
                                        written, as for a miss */
#define NNBVH_INTERACTION_PATCH 3    /* all fields valid (BilinearPatch::InteractionFromIntersection,
                                        shapes.h:1396-1489) */
typedef struct nnbvh_interaction {   /* 192 B */
    float pi_lo[3], pi_hi[3]; /* Point3fi pi = pHit +- gamma(7) sum|b_i p_i| (shapes.h:926-930) */
    float uv[2];
    float wo[3];              /* Normalize(-ray.d) */
    float time;
    float n[3];               /* geometric normal after FaceForward to the shading normal */
    int32_t face_index;
    float dpdu[3], dpdv[3];
    float ns[3], dpdus[3], dpdvs[3], dndus[3], dndvs[3]; /* SurfaceInteraction::shading */
    float dndu[3], dndv[3];   /* geometric normal derivatives (zero for triangles) */
    float pad0;
    int32_t prim;             /* the hit record's primitive id, -1 = miss */
    int32_t status;           /* NNBVH_INTERACTION_* */
    int32_t pad1[2];
} nnbvh_interaction;
/* One of d_rays (nnbvh_ray[max_items]) / ray_soa gives the rays the hits belong to (direction and
 * time are read).  n = min(max_items, *d_size) when d_size != NULL.  d_out: nnbvh_interaction
 * [max_items], record i for hit i. */
/* host buffers (synchronous; copies in and out, like nnbvh_intersect_closest) */
int nnbvh_triangle_interactions(const nnbvh_shading_mesh *m, const nnbvh_ray *rays, const nnbvh_hit *hits,
                                int32_t n, nnbvh_interaction *out);
int nnbvh_triangle_interactions_device(const nnbvh_shading_mesh *m, const void *d_rays,
                                       const nnbvh_ray_soa *ray_soa, const void *d_hits,
                                       int32_t max_items, const int32_t *d_size, void *d_out,
                                       void *stream);

/* ---- KdTreeAggregate (cpu/aggregates.h:75-105; aggregates.cpp:746-1161) --------------------------
 * The reference's other accelerator ("kdtree" in CreateAccelerator, aggregates.cpp:1163-1178) and the
 * native structure of the learned trees of machine_learning/nss_*.py.
 *   nnbvh_kd_node             <- KdTreeNode (aggregates.cpp:753-775): 8 B; interior = {Float split,
 *                                flags = axis | aboveChild << 2} with the below child at index + 1; leaf =
 *                                {onePrimitiveIndex | primitiveIndicesOffset, flags = 3 | nPrimitives << 2}
 *   nnbvh_kd_build_create     <- KdTreeAggregate ctor + buildTree (aggregates.cpp:798-971), host-side
 *   nnbvh_kd_scene_create     <- takes what KdTreeAggregate owns after construction: nodes,
 *                                primitiveIndices, the primitives in their ORIGINAL order, bounds
 *   nnbvh_kd_intersect_*      <- KdTreeAggregate::Intersect / IntersectP (aggregates.cpp:973-1150);
 *                                nodes_visited = kdNodesVisited (aggregates.cpp:796), prim_tests =
 *                                nTriTests; a primitive that overlaps several leaves is tested in each
 *                                (the reference has no mailboxing). */
typedef struct nnbvh_kd_node {
    uint32_t split_or_index; /* bit pattern of the float split position, or the leaf's int32 */
    uint32_t flags;
} nnbvh_kd_node;
typedef struct nnbvh_kd_build nnbvh_kd_build;
typedef struct nnbvh_kd_scene nnbvh_kd_scene;
/* defaults of KdTreeAggregate::Create (aggregates.cpp:1152-1161): isect_cost 5, traversal_cost 1,
 * empty_bonus 0.5, max_prims 1, max_depth -1 (= round(8 + 1.3 log2 n)).  prim_bounds as for
 * nnbvh_build_create_with_bounds (read for NNBVH_PRIM_HOST entries).  Triangles, patches and
 * host-only primitives; instances are not accepted. */
nnbvh_kd_build *nnbvh_kd_build_create(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                      const float *prim_bounds, int isect_cost, int traversal_cost,
                                      float empty_bonus, int max_prims, int max_depth);
/* The same construction on the GPU (`device`), level by level (kd_build_gpu.hip): the identical node array —
 * split planes, child links, leaf sizes and primitiveIndices offsets do not depend on how a sort orders EQUAL
 * (t, type) edges — with the primitives inside a multi-primitive leaf in std::stable_sort order, where
 * nnbvh_kd_build_create leaves them as libstdc++'s std::sort does (aggregates.cpp:899-903 leaves that order
 * to the standard library).  nnbvh_kd_build_create_stable is the host builder with that same order: the
 * device builder's byte-for-byte checker.  No CPU fallback: without a HIP device the call fails. */
nnbvh_kd_build *nnbvh_kd_build_create_gpu(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                          const float *prim_bounds, int isect_cost, int traversal_cost,
                                          float empty_bonus, int max_prims, int max_depth, int device);
nnbvh_kd_build *nnbvh_kd_build_create_stable(const nnbvh_prim *prims, int n_prims, const float *verts, int n_verts,
                                             const float *prim_bounds, int isect_cost, int traversal_cost,
                                             float empty_bonus, int max_prims, int max_depth);
/* milliseconds of the device builder: [0] on the device (upload of the bounds .. both arrays written),
 * [1] including the download; zeros for a host build */
int nnbvh_kd_build_timing(const nnbvh_kd_build *b, double out_ms[2]);
const nnbvh_kd_node *nnbvh_kd_build_nodes(const nnbvh_kd_build *b, int *n_nodes);
const int32_t *nnbvh_kd_build_prim_indices(const nnbvh_kd_build *b, int *n_indices);
int nnbvh_kd_build_bounds(const nnbvh_kd_build *b, float out_min_max[6]);
int nnbvh_kd_build_depth(const nnbvh_kd_build *b); /* edges root -> deepest node */
void nnbvh_kd_build_destroy(nnbvh_kd_build *b);

/* The tree is validated (child links, leaf ranges, primitive indices, depth <= 64 = the reference's
 * toVisit[64], aggregates.cpp:982) before anything is uploaded.  prims: n_prims primitives indexed
 * by the leaves (triangles, alpha-tested triangles NNBVH_PRIM_ALPHA_TRIANGLE[_FLIPPED], bilinear
 * patches, host-only primitives); hit.prim reports nnbvh_prim.id. */
nnbvh_kd_scene *nnbvh_kd_scene_create(const nnbvh_kd_node *nodes, int n_nodes, const int32_t *prim_indices,
                                      int n_indices, const nnbvh_prim *prims, int n_prims,
                                      const float *verts, int n_verts, const float bounds_min_max[6],
                                      int device);
/* ... with the attributes the alpha-tested kinds read: normals and uvs per vertex, prim_alpha per entry of `prims`
 * (kd primitives are in the caller's order).  A kind whose arrays are missing stays the host's (record void), as
 * every such kind does through nnbvh_kd_scene_create. */
nnbvh_kd_scene *nnbvh_kd_scene_create_with_attributes(const nnbvh_kd_node *nodes, int n_nodes,
                                                      const int32_t *prim_indices, int n_indices,
                                                      const nnbvh_prim *prims, int n_prims, const float *verts,
                                                      int n_verts, const float bounds_min_max[6], const float *normals,
                                                      const float *uvs, const float *prim_alpha, int device);
void nnbvh_kd_scene_destroy(nnbvh_kd_scene *s);
/* host buffers (synchronous) */
int nnbvh_kd_intersect_closest(nnbvh_kd_scene *s, const nnbvh_ray *rays, int64_t n, nnbvh_hit *hits);
int nnbvh_kd_intersect_any(nnbvh_kd_scene *s, const nnbvh_ray *rays, int64_t n, uint8_t *occluded,
                           int32_t *nodes_visited, int32_t *prim_tests);
/* device buffers, asynchronous on `stream` */
int nnbvh_kd_intersect_closest_device(nnbvh_kd_scene *s, const void *d_rays, int64_t n, void *d_hits,
                                      void *stream);
int nnbvh_kd_intersect_any_device(nnbvh_kd_scene *s, const void *d_rays, int64_t n, void *d_occluded,
                                  void *d_nodes_visited, void *d_prim_tests, void *stream);

/* tuning knobs (speed only, never results): "stack_window" (LDS entries per lane: 4, 8, 16),
 * "blocks_per_cu" (0 = auto), "xcd_queues" (0/1), "prim_weight" / "refill_weight" (1..64: how much a
 * lane waiting on a primitive test / an idle lane counts against a lane waiting on an interior node,
 * which counts 16, when a wavefront picks its next step), "int_repeat" / "prim_repeat" (1..16
 * interior / primitive steps per scheduling decision; the kernel instances for scenes without patches,
 * instances and host-only primitives always take one primitive step), "fused_batches" (0/1: nnbvh_trace_batches_device as one launch where the
 * batches allow it).  Returns NNBVH_ERR_ARG for unknown keys. */
int nnbvh_scene_set_option(nnbvh_scene *s, const char *key, int value);

/* diagnostics: wavefront scheduling statistics accumulated since the last reset —
 * out = {interior trips, interior lanes, primitive trips, primitive lanes, refill trips,
 * refill lanes, and over the interior trips the sums of lanes waiting on an interior node,
 * on a primitive, idle; one spare; shader cycles spent in interior / primitive / refill trips; one
 * spare; interior steps executed and the lanes that took part}.  All zero unless built with
 * -DNNBVH_STATS. */
int nnbvh_scene_sched_stats(nnbvh_scene *s, uint64_t out[16], int reset);

#ifdef __cplusplus
}
#endif
#endif
