/*
 * nnbvh_oracle.h — TEST INFRASTRUCTURE ONLY (the parity oracle).
 *
 * Plain-C CPU restatement of the reference's BVH traversal hot path.  Nothing in
 * the product (nn_bvh_amd/, include/) may include, link or call this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as
 * the checker / the reported CPU baseline.
 *
 * Wire structs are byte-identical to include/nnbvh.h so the same buffers feed
 * both sides; they are re-declared here so the oracle has no dependency on the
 * product headers.
 */
#ifndef NNBVH_ORACLE_H
#define NNBVH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == LinearBVHNode, /root/reference/src/pbrt/cpu/aggregates.cpp:129-137 (32 B, alignas 32) */
typedef struct {
    float pmin[3];
    float pmax[3];
    int32_t offset;   /* leaf: primitivesOffset; interior: secondChildOffset */
    uint16_t nprims;  /* 0 -> interior */
    uint8_t axis;
    uint8_t pad;
} orc_node;

/* One entry of the leaf-ordered `primitives` vector (aggregates.cpp:176): the
 * Triangle{meshIndex,triIndex} / BilinearPatch{meshIndex,blpIndex} handle
 * (shapes.h:1188, 1535) flattened to global vertex indices. */
typedef struct {
    int32_t kind; /* 0 = triangle (v[0..2]), 1 = bilinear patch (v = p00,p10,p01,p11), 2 = instance (v[0]), 3 = host-only */
    int32_t id;   /* caller's original primitive index, returned on a hit */
    int32_t v[4];
} orc_prim;

/* TransformedPrimitive (cpu/primitive.h:148-172): a child BVHAggregate behind a static
 * render-from-primitive transform.  A top-level primitive of kind 2 names one by v[0]. */
typedef struct {
    float m[12];      /* renderFromPrimitive, rows 0..2 of the 4x4 (row 3 = 0 0 0 1) */
    float m_inv[12];  /* its inverse, same layout */
    int32_t root;     /* index of the child tree's root in the shared node array */
    int32_t n_nodes;  /* nodes of the child tree (contiguous from root) */
} orc_instance;

typedef struct {
    float o[3];
    float tmax;
    float d[3];
    float time;
} orc_ray;

typedef struct {
    int32_t prim; /* -1 = miss */
    float t;
    float b0, b1, b2; /* triangle: barycentrics; patch: b0=u, b1=v, b2=0 */
    int32_t nodes_visited;
    int32_t prim_tests;
    int32_t instance; /* -1 = reached a host-only primitive (record void); 0 = hit in the top level
                         (or miss); k+1 = hit inside instance k */
} orc_hit;

/* leaf tests and slab test on single inputs (return 1 = hit) */
int orc_slab(const float bounds[6], const float o[3], const float d[3], float ray_tmax);
int orc_triangle(const float o[3], const float d[3], float tmax, const float p0[3],
                 const float p1[3], const float p2[3], float out_b0b1b2t[4]);
int orc_bilinear_patch(const float o[3], const float d[3], float tmax, const float p00[3],
                       const float p10[3], const float p01[3], const float p11[3],
                       float out_uvt[3]);

/* batched forms used to cross-check against oracle/_ref/ref_leaf */
void orc_slab_batch(const float *bounds6, const float *o3, const float *d3,
                    const float *tmax, int n, uint8_t *out);
void orc_triangle_batch(const float *o3, const float *d3, const float *tmax,
                        const float *p9, int n, uint8_t *hit, float *out4);
void orc_bilinear_patch_batch(const float *o3, const float *d3, const float *tmax,
                              const float *p12, int n, uint8_t *hit, float *out3);

/* BVHAggregate::Intersect / IntersectP over a ray batch; nthreads<=1 = serial */
void orc_intersect_closest(const orc_node *nodes, int n_nodes, const orc_prim *prims,
                           const float *verts, const orc_ray *rays, int64_t n,
                           orc_hit *hits, int nthreads);
void orc_intersect_any(const orc_node *nodes, int n_nodes, const orc_prim *prims,
                       const float *verts, const orc_ray *rays, int64_t n,
                       uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests,
                       int nthreads);

/* Transform::ApplyInverse(const Ray&, Float *tMax) (util/transform.h:416-429) for an affine
 * transform given by the 3x4 inverse matrix: out = o'[3], d'[3], tmax'. */
void orc_apply_inverse_ray(const float m_inv[12], const float o[3], const float d[3], float tmax,
                           float out7[7]);
void orc_apply_inverse_ray_batch(const float *m_inv12, const float *o3, const float *d3,
                                 const float *tmax, int n, float *out7);
/* Transform::operator()(const Bounds3f&) (util/transform.cpp): bounds of the 8 transformed corners */
void orc_transform_bounds(const float m[12], const float in6[6], float out6[6]);

/* two-level forms: prims of kind 2 refer to `instances` (NULL = none) */
void orc_intersect_closest_inst(const orc_node *nodes, const orc_prim *prims, const float *verts,
                                const orc_instance *instances, const orc_ray *rays, int64_t n,
                                orc_hit *hits, int nthreads);
void orc_intersect_any_inst(const orc_node *nodes, const orc_prim *prims, const float *verts,
                            const orc_instance *instances, const orc_ray *rays, int64_t n,
                            uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests,
                            int nthreads);

/* wavefront queue rules (wavefront/intersect.h:16-156) on work-item indices; queues / sizes in
 * the order escaped, hit_area_light, basic_eval, universal_eval, medium_sample, next_ray */
void orc_wavefront_enqueue_closest(const orc_hit *hits, int n, const uint8_t *has_medium,
                                   const uint8_t *prim_class, int64_t n_class,
                                   int32_t *const queues[6], int32_t sizes[6]);
/* RecordShadowRayResult (wavefront/intersect.h:32-47); Ld, r_u, r_l, L: 4 floats per item/pixel */
void orc_record_shadow(const uint8_t *occluded, int n, const float *Ld, const float *r_u,
                       const float *r_l, const int32_t *pixel_index, float *L);

/* == KdTreeNode, cpu/aggregates.cpp:753-775 (8 B): interior {float split; flags = axis | above << 2},
 * leaf {int onePrimitiveIndex | primitiveIndicesOffset; flags = 3 | nPrimitives << 2} */
typedef struct {
    uint32_t split_or_index;
    uint32_t flags;
} orc_kd_node;
/* Bounds3::IntersectP(o, d, tMax, *hitt0, *hitt1), util/vecmath.h:1547-1571: the entry / exit
 * distances KdTreeAggregate starts from.  Returns 1 = hit and writes t0t1[2]. */
int orc_bounds_t0t1(const float bounds[6], const float o[3], const float d[3], float tmax, float t0t1[2]);
void orc_bounds_t0t1_batch(const float *bounds6, const float *o3, const float *d3, const float *tmax, int n,
                           uint8_t *hit, float *t0t1);
/* KdTreeAggregate::Intersect / IntersectP (aggregates.cpp:973-1150).  prims are indexed by the leaves
 * (original order); hits carry nodes_visited = kdNodesVisited and prim_tests = nTriTests. */
void orc_kd_intersect_closest(const orc_kd_node *nodes, const int32_t *prim_indices, const orc_prim *prims,
                              const float *verts, const float bounds[6], const orc_ray *rays, int64_t n,
                              orc_hit *hits, int nthreads);
void orc_kd_intersect_any(const orc_kd_node *nodes, const int32_t *prim_indices, const orc_prim *prims,
                          const float *verts, const float bounds[6], const orc_ray *rays, int64_t n,
                          uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests, int nthreads);

/* util/hash.h:19-128: MurmurHash64A over the bytes of (a[3], b[3]) = Hash(Point3f, Vector3f) =
 * Hash(Point3f, Point3f); HashFloat = uint32(hash) * 2^-32 */
uint64_t orc_hash_6f(const float a[3], const float b[3]);
float orc_hash_float_6f(const float a[3], const float b[3]);
/* OffsetRayOrigin / SpawnRayTo (ray.h:75-101) on a Point3fi given as lo[3], hi[3] */
void orc_offset_ray_origin(const float pi_lo[3], const float pi_hi[3], const float n[3], const float w[3],
                           float out[3]);
void orc_spawn_ray_to(const float pi_lo[3], const float pi_hi[3], const float n[3], const float p_to[3],
                      float out_o[3], float out_d[3]);
/* WeightedReservoirSampler<int> seeded with `seed` (util/sampling.h:524-596 on the PCG32 RNG of
 * util/rng.h): n_adds candidates 0..n-1 of weight 1; returns the selected index (-1 = none) */
int orc_wrs_unit_weights(uint64_t seed, int n_adds, float *sample_probability, float *weight_sum);
/* batch drivers for the cross-checks against oracle/_ref/ref_leaf */
void orc_hash_batch(const float *in6, int n, uint32_t *lo, uint32_t *hi, float *hash_float);
void orc_offset_batch(const float *in12, int n, float *out9);
void orc_wrs_batch(const float *in7, int n, int32_t *selected, float *out2);

/* AnimatedTransform's members (util/transform.h:443-520) as the reference object holds them after
 * construction, and AnimatedTransform::Interpolate (util/transform.cpp:1062-1081): Translate(lerp T) *
 * Transform(Slerp(R)) * Transform(lerp S) with Transform::operator* (FMA chains, math.h:1499-1509) and the
 * 4x4 Inverse of math.h:1572-1625.  Matrices are full 4x4, row-major. */
typedef struct {
    float start_m[16], start_minv[16], end_m[16], end_minv[16];
    float T[2][3];
    float R[2][4]; /* v.x v.y v.z w */
    float S[2][16];
    float start_time, end_time;
    int32_t actually_animated;
    int32_t pad;
} orc_anim;
void orc_anim_interpolate(const orc_anim *a, float time, float m[16], float minv[16]);
/* 0 (default): Slerp's per-ray sines with libm's sinf, as the reference; 1: evaluated in double and
 * rounded once, as the device does (its documented tolerance exception) */
void orc_set_sin_mode(int mode);
/* per-vertex shading normals (3 floats per vertex, indexed like verts) for prim kinds 6 / 7 (alpha-tested
 * triangles of smooth meshes); NULL = none.  The pointer is kept: it must outlive the traces that use it. */
void orc_set_vertex_normals(const float *normals);
/* constant alpha per primitive (one float per entry of the prims array handed to orc_intersect_closest / _any)
 * for prim kinds 8 .. 15 (alpha-tested bilinear patches: 8 + flipped + 2 * smooth + 4 * uv); NULL = none */
void orc_set_prim_alpha(const float *alpha);
/* (u, v) per vertex (2 floats, indexed like verts) for prim kinds 12 .. 15; NULL = none */
void orc_set_vertex_uvs(const float *uvs);
void orc_anim_interpolate_batch(const orc_anim *a, const float *time, int n, float *out32);
/* two-level traversal with AnimatedPrimitive instances (cpu/primitive.cpp:133-158): anims[k] belongs to
 * instances[k]; entries with actually_animated == 0 are TransformedPrimitives */
void orc_intersect_closest_anim(const orc_node *nodes, const orc_prim *prims, const float *verts,
                                const orc_instance *instances, const orc_anim *anims, const orc_ray *rays,
                                int64_t n, orc_hit *hits, int nthreads);
void orc_intersect_any_anim(const orc_node *nodes, const orc_prim *prims, const float *verts,
                            const orc_instance *instances, const orc_anim *anims, const orc_ray *rays,
                            int64_t n, uint8_t *occluded, int32_t *nodes_visited, int32_t *prim_tests,
                            int nthreads);

/* UpdateFilm + RGBFilm::AddSample (wavefront/film.cpp:13-40, film.h:239-255) without the spectral
 * sensor conversion: sample (pass, i) of pixel slot i adds weight * clamp(rgb) to pixels[4 * pixel]
 * (double rgbSum[3], weightSum; film.h:302-307), passes in order.  bounds = x0 y0 x1 y1. */
void orc_film_add_samples(double *pixels, const int32_t bounds[4], float max_component,
                          const int32_t *px, const int32_t *py, const float *rgb, int rgb_stride,
                          const float *weight, int n_per_pass, int n_passes);

/* Triangle::InteractionFromIntersection (shapes.h:884-1010): 44-float record, layout in the .c;
 * uv6 / n9 / s9 NULL = mesh without that attribute */
int orc_triangle_interaction(const float p9[9], const float *uv6, const float *n9, const float *s9,
                             int flip_normal, const float b[3], const float wo[3], float time,
                             int face_index, float out[44]);
void orc_triangle_interaction_batch(const float *in45, int n, float *out44);

/* BilinearPatch::InteractionFromIntersection (shapes.h:1396-1489): 50-float record = the 44 above +
 * geometric dndu[3], dndv[3]; p12 = p00 p10 p01 p11 */
int orc_patch_interaction(const float p12[12], const float *uv8, const float *n12, int flip_normal,
                          const float hit_uv[2], const float wo[3], float time, int face_index,
                          float out[50]);
void orc_patch_interaction_batch(const float *in40, int n, float *out50);

/* Transform::operator()(const SurfaceInteraction &) (util/transform.cpp:229-261), affine 3x4 m / m_inv;
 * fields: pi low[3] high[3], n, wo, dpdu, dpdv, dndu, dndv, shading n, dpdu, dpdv, dndu, dndv */
void orc_transform_interaction(const float m[12], const float m_inv[12], const float in39[39], float out39[39]);
void orc_transform_interaction_batch(const float *in72, int n, float *out40);

/* brute force closest hit over all prims in index order (no BVH): a second,
 * tree-independent check of t for the traversal restatement. */
void orc_brute_closest(const orc_prim *prims, int n_prims, const float *verts,
                       const orc_ray *rays, int64_t n, orc_hit *hits);

#ifdef __cplusplus
}
#endif
#endif
