/* trace_demo.c — the C ABI from plain C (no C++, no Python): build a tree over a few
 * triangles on the host, upload it, trace a closest-hit and an any-hit batch.
 *
 *   gcc -std=c11 -Iinclude examples/trace_demo.c -Lnn_bvh_amd -l:libnnbvh_hip.so \
 *       -Wl,-rpath,$PWD/nn_bvh_amd -o trace_demo && ./trace_demo
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "nnbvh.h"

int main(void) {
    /* a 32 x 32 grid of triangles in the plane z = 0 */
    enum { N = 32 };
    int n_verts = (N + 1) * (N + 1), n_tris = 2 * N * N;
    float *verts = malloc(sizeof(float) * 3 * (size_t)n_verts);
    nnbvh_prim *prims = malloc(sizeof(nnbvh_prim) * (size_t)n_tris);
    for (int i = 0; i <= N; ++i)
        for (int j = 0; j <= N; ++j) {
            float *v = &verts[3 * (i * (N + 1) + j)];
            v[0] = (float)i;
            v[1] = (float)j;
            v[2] = 0.0f;
        }
    for (int i = 0, k = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            int a = i * (N + 1) + j, b = a + N + 1;
            prims[k] = (nnbvh_prim){NNBVH_PRIM_TRIANGLE, k, {a, b, b + 1, 0}};
            ++k;
            prims[k] = (nnbvh_prim){NNBVH_PRIM_TRIANGLE, k, {a, b + 1, a + 1, 0}};
            ++k;
        }
    nnbvh_build *b = nnbvh_build_create(prims, n_tris, verts, n_verts, 4, NNBVH_SPLIT_SAH);
    if (!b) {
        fprintf(stderr, "build: %s\n", nnbvh_last_error());
        return 1;
    }
    int n_nodes = 0, n_ordered = 0;
    const nnbvh_linear_node *nodes = nnbvh_build_nodes(b, &n_nodes);
    const nnbvh_prim *ordered = nnbvh_build_ordered_prims(b, &n_ordered);
    printf("%d triangles -> %d nodes, depth %d\n", n_tris, n_nodes, nnbvh_build_depth(b));
    if (nnbvh_device_count() < 1) {
        printf("no HIP device: host side only (the traversal has no CPU fallback)\n");
        nnbvh_build_destroy(b);
        return 0;
    }
    nnbvh_scene *s = nnbvh_scene_create(nodes, n_nodes, ordered, n_ordered, verts, n_verts, 0);
    nnbvh_build_destroy(b);
    if (!s) {
        fprintf(stderr, "scene: %s\n", nnbvh_last_error());
        return 1;
    }
    enum { R = 4 };
    nnbvh_ray rays[R] = {
        {{3.25f, 7.5f, 5.0f}, INFINITY, {0, 0, -1}, 0},        /* straight down: hits at t = 5 */
        {{3.25f, 7.5f, 5.0f}, 4.0f, {0, 0, -1}, 0},            /* tMax cuts it off */
        {{-3.0f, 7.5f, 5.0f}, INFINITY, {0, 0, -1}, 0},        /* beside the grid */
        {{0.5f, 0.5f, 2.0f}, 1.0f - 1e-4f, {20.0f, 20.0f, -4.0f}, 0}, /* shadow-style, un-normalised */
    };
    nnbvh_hit hits[R];
    unsigned char occ[R];
    if (nnbvh_intersect_closest(s, rays, R, hits) || nnbvh_intersect_any(s, rays, R, occ, NULL, NULL)) {
        fprintf(stderr, "trace: %s\n", nnbvh_last_error());
        return 1;
    }
    for (int i = 0; i < R; ++i)
        printf("ray %d: prim %d t %g b (%g %g %g) nodes visited %d, occluded %d\n", i, hits[i].prim,
               hits[i].t, hits[i].b0, hits[i].b1, hits[i].b2, hits[i].nodes_visited, occ[i]);
    nnbvh_scene_destroy(s);
    free(verts);
    free(prims);
    return !(hits[0].prim >= 0 && hits[0].t == 5.0f && hits[1].prim < 0 && hits[2].prim < 0 && occ[0] == 1);
}
