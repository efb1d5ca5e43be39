"""CPU tests of the oracle itself: pinned bit-for-bit to vectors generated from the REFERENCE's
own compiled leaf functions (tests/golden/leaf_*.npz, tests/golden/make_leaf_golden.py), the
reference's Triangle.BadCases known-answer case, and tree-independent cross-checks of the
restated traversal loop."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import build_tree, scene

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("mode", ["tri", "blp", "slab"])
def test_leaf_functions_match_reference_vectors_bit_exact(mode):
    g = np.load(os.path.join(GOLD, f"leaf_{mode}.npz"))
    hit, out = ob.leaf_batch(mode, g["inputs"])
    assert (hit == g["hit"]).all()
    h = g["hit"].astype(bool)
    assert (out.view(np.uint32)[h] == g["out_bits"][h]).all()
    assert 0.2 < h.mean() < 0.9  # both outcomes are exercised


def test_ray_inverse_transform_matches_reference_vectors_bit_exact():
    """Transform::ApplyInverse(Ray, tMax) (util/transform.h:416-429), the ray transform of
    TransformedPrimitive, against vectors from the reference binary."""
    g = np.load(os.path.join(GOLD, "leaf_xfray.npz"))
    r = g["inputs"]
    out = ob.apply_inverse_ray(r[:, 23:35], r[:, 0:3], r[:, 3:6], r[:, 6])
    assert (out.view(np.uint32) == g["out_bits"]).all()
    # the origin really moves (error-bound offset) on most inputs
    assert (out[:, 6] != r[:, 6]).mean() > 0.2


def test_transform_bounds_match_reference_vectors_bit_exact():
    """Transform::operator()(Bounds3f) (util/transform.cpp:134-139) = TransformedPrimitive::Bounds():
    the oracle's restatement AND the product's host function (nnbvh_transform_bounds, used when a
    two-level scene is assembled) against vectors from the reference binary."""
    from nn_bvh_amd.instancing import transform_bounds
    g = np.load(os.path.join(GOLD, "leaf_xfbounds.npz"))
    r = g["inputs"]
    for i in range(len(r)):
        exp = g["out_bits"][i]
        assert (ob.transform_bounds(r[i, 0:12], r[i, 16:22]).view(np.uint32) == exp).all(), i
        got = np.asarray(transform_bounds(r[i, 0:12], r[i, 16:22]), np.float32)
        assert (got.view(np.uint32) == exp).all(), i


def test_hlbvh_leaf_order_follows_the_reference_morton_codes(nnbvh_lib):
    """buildHLBVH's per-primitive Morton codes (Bounds3::Offset + EncodeMorton3, aggregates.cpp:398-408)
    computed by the reference binary for a 6 000-triangle soup; HLBVH's leaf-ordered primitive table
    is the primitives in stable-sorted code order (the 5 x 6-bit LSD radix sort is stable), which pins
    the host builder's codes, its sort, and — through tests/test_gpu_build.py — the device builder's."""
    from nn_bvh_amd import build_tree, make_prims
    g = np.load(os.path.join(GOLD, "hlbvh_morton.npz"))
    verts, codes = g["verts"], g["codes"]
    n = len(codes)
    prims = make_prims(np.arange(3 * n, dtype=np.int32).reshape(n, 3))
    tree = build_tree(prims, verts, 4, "hlbvh")
    assert np.array_equal(tree.ordered_prims["id"], np.argsort(codes, kind="stable"))
    assert len(np.unique(codes)) > 4000 and (np.bincount(np.unique(codes, return_inverse=True)[1]) >= 200).any()
    assert (codes < (1 << 30)).all()


def test_golden_vectors_reach_the_rare_branches():
    """The vectors must include exact-zero edge functions (the fp64 fallback), degenerate
    triangles and zero direction components, or the pin would not cover those branches."""
    g = np.load(os.path.join(GOLD, "leaf_tri.npz"))
    r = g["inputs"]
    # the first 256 cases aim at integer vertices / edge midpoints: hits with a barycentric of 0
    bary = g["out_bits"][:256, :3].view(np.float32)
    assert ((bary == 0).any(1) & (g["hit"][:256] == 1)).sum() > 20
    p = r[:, 7:].reshape(-1, 3, 3)
    assert ((p[:, 1] == p[:, 2]).all(1)).sum() > 100       # degenerate triangles
    assert ((r[:, 3:6] == 0).sum(1) == 2).sum() > 100       # axis-aligned directions
    s = np.load(os.path.join(GOLD, "leaf_slab.npz"))["inputs"]
    assert ((s[:, 3:6] == 0).any(1)).sum() > 500            # +-inf inverse directions


def test_triangle_badcases_known_answer():
    """/root/reference/src/pbrt/shapes_test.cpp:435-449: this ray must miss this triangle."""
    rec = np.array([[-1081.47925, 99.9999542, 87.7701111, -32.1072998, -183.355865, -144.607635,
                     np.inf, -1113.45459, -79.049614, -56.2431908, -1113.45459, -87.0922699,
                     -56.2431908, -1113.45459, -79.2090149, -56.2431908]], np.float32)
    hit, _ = ob.leaf_batch("tri", rec)
    assert hit[0] == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_traversal_agrees_with_brute_force(seed):
    """BVH closest hit == closest hit over ALL primitives (no tree): same t always; same
    primitive unless two primitives tie at that t."""
    verts, prims = ss.random_soup(400, 100, seed)
    tree = build_tree(prims, verts)
    rays = np.concatenate([scene.random_rays(1500, verts.min(0) - 1, verts.max(0) + 1, seed),
                           ss.edge_case_rays(verts, prims, seed, 1024)])
    # A zero direction component makes invDir infinite, and a box face through the ray's plane
    # then yields 0 * inf = NaN in the slab test (vecmath.h:1587-1607): the reference's BVH
    # misses such boxes although the triangle test alone would hit.  That behaviour is part of
    # the contract (the GPU reproduces it); a tree-free brute force cannot, so skip those rays.
    rays = rays[(rays["d"] != 0).all(1)]
    h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
    b = ob.brute_closest(prims, verts, rays)
    assert ((h["prim"] >= 0) == (b["prim"] >= 0)).all()
    assert (h["t"].view(np.uint32) == b["t"].view(np.uint32)).all()
    assert (h["prim"] != b["prim"]).mean() < 0.01
    occ, vis, _ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays)
    assert (occ == (b["prim"] >= 0)).all()
    assert (vis <= h["nodes_visited"]).all()  # any-hit stops early, never visits more


def test_traversal_golden_fixture_is_reproduced():
    g = np.load(os.path.join(GOLD, "traversal_small.npz"))
    h = ob.closest(g["nodes"], g["ordered_prims"], g["verts"], g["rays"])
    assert h.tobytes() == g["hits"].tobytes()
    occ, vis, tst = ob.any_hit(g["nodes"], g["ordered_prims"], g["verts"], g["rays"])
    assert (occ == g["occ"]).all() and (vis == g["occ_visited"]).all() and (tst == g["occ_tests"]).all()


def test_threaded_oracle_equals_serial():
    g = np.load(os.path.join(GOLD, "traversal_small.npz"))
    a = ob.closest(g["nodes"], g["ordered_prims"], g["verts"], g["rays"], nthreads=1)
    b = ob.closest(g["nodes"], g["ordered_prims"], g["verts"], g["rays"], nthreads=4)
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("name,nodes,depth,v_closest,t_closest,v_any,t_any", [
    ("killeroos", 129771, 24, 19.8, 2.57, 16.1, 1.69),
    ("coffee_maker", 321163, 28, 36.0, 2.58, 28.5, 1.83),
    ("bathroom", 1033239, 32, 63.0, 4.18, 40.7, 1.99),
    ("crown", 6462477, 39, 99.4, 5.05, 78.8, 3.81),
])
def test_reference_aggregates_recorded_in_survey(name, nodes, depth, v_closest, t_closest, v_any,
                                                 t_any):
    """Outputs of the REFERENCE BVHAggregate recorded in SURVEY.md §6 / BASELINE.md §2 for the
    scenes' pixel-centre primary rays: node count, tree depth, mean nodes visited (V) and mean
    triangle tests (T), closest and any hit.  Our builder + oracle traversal must reproduce them
    on ALL pixel centres to the digits recorded (half a unit of the last recorded digit).  This is
    the strongest reference anchor the traversal loop has (BVHAggregate itself cannot be built
    here: DESIGN.md §2); per-ray parity with the reference stays unpinned.  The survey's bounce-ray
    rows are not used: its bounce generator is not specified to the bit, and ours (same recipe,
    own random stream) gives crown V = 93.0 / T = 7.73 against the recorded 89.4 / 6.95.
    Needs the git-ignored scene blob (build container / GPU box)."""
    if not os.path.exists(scene.blob_path(name)):
        pytest.skip(f"data/{name}.npz not present")
    from nn_bvh_amd import make_prims
    verts, tris = scene.load_blob(name)
    tree = build_tree(make_prims(tris), verts)
    assert len(tree.nodes) == nodes and tree.depth == depth
    rays = scene.camera_rays(name, jitter=False)
    nthreads = min(8, os.cpu_count() or 1)
    h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads)

    def agrees(measured, recorded):
        digits = len(repr(recorded).split(".")[1])
        return abs(measured - recorded) <= 0.5 * 10.0 ** -digits

    assert agrees(h["nodes_visited"].mean(), v_closest), h["nodes_visited"].mean()
    assert agrees(h["prim_tests"].mean(), t_closest), h["prim_tests"].mean()
    _, vis, tst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, nthreads)
    assert agrees(vis.mean(), v_any), vis.mean()
    assert agrees(tst.mean(), t_any), tst.mean()
