"""CPU tests of the host-side pieces: the SAH builder's invariants, the C-ABI surface
(every symbol include/nnbvh.h declares is exported) and its error behaviour without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import scenes_small as ss
from nn_bvh_amd import NODE_DTYPE, PRIM_DTYPE, NNBVHError, _lib, build_tree, make_prims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_tree(tree, prims, verts, max_prims):
    nodes, ordered = tree.nodes, tree.ordered_prims
    assert sorted(ordered["id"].tolist()) == sorted(prims["id"].tolist())  # a permutation
    n = len(nodes)
    assert n == 2 * int((nodes["nprims"] > 0).sum()) - 1
    covered = np.zeros(len(ordered), int)

    def walk(i, depth):
        nd = nodes[i]
        if nd["nprims"] > 0:
            covered[nd["offset"]: nd["offset"] + nd["nprims"]] += 1
            pv = verts[ordered["v"][nd["offset"]: nd["offset"] + nd["nprims"]]]
            ok = np.where(ordered["kind"][nd["offset"]: nd["offset"] + nd["nprims"], None, None] == 0,
                          np.arange(4)[None, :, None] < 3, True)
            lo = np.where(ok, pv, np.inf).min((0, 1))
            hi = np.where(ok, pv, -np.inf).max((0, 1))
            assert (lo == nd["pmin"]).all() and (hi == nd["pmax"]).all()  # tight leaf bounds
            return i + 1, depth
        a, b = i + 1, nd["offset"]
        assert a < b < n and nd["axis"] in (0, 1, 2)
        assert (np.minimum(nodes[a]["pmin"], nodes[b]["pmin"]) == nd["pmin"]).all()
        assert (np.maximum(nodes[a]["pmax"], nodes[b]["pmax"]) == nd["pmax"]).all()
        end_a, da = walk(a, depth + 1)
        assert end_a == b  # DFS layout: second child follows the first child's subtree
        end_b, db = walk(b, depth + 1)
        return end_b, max(da, db)

    end, depth = walk(0, 0)
    assert end == n and depth == tree.depth
    assert (covered == 1).all()


@pytest.mark.parametrize("split,max_prims", [("sah", 4), ("sah", 1), ("middle", 4), ("equal", 2),
                                             ("hlbvh", 4), ("hlbvh", 1)])
def test_builder_invariants(nnbvh_lib, split, max_prims):
    verts, prims = ss.random_soup(700, 150, 4)
    tree = build_tree(prims, verts, max_prims, split)
    check_tree(tree, prims, verts, max_prims)
    if split == "hlbvh":
        # emitLBVH makes a leaf when nPrimitives < maxPrimsInNode (strictly), aggregates.cpp:459
        leaves = tree.nodes[tree.nodes["nprims"] > 0]
        assert len(leaves) > 10
    if split == "sah" and max_prims == 4:
        # SAH never makes a leaf above maxnodeprims unless centroids coincide (aggregates.cpp:337)
        assert tree.nodes["nprims"].max() <= 4


def test_builder_big_leaf_on_coincident_centroids(nnbvh_lib):
    verts, prims = ss.coincident_centroids(300, 2)
    tree = build_tree(prims, verts)
    check_tree(tree, prims, verts, 4)
    assert tree.nodes["nprims"].max() > 4  # bypasses maxnodeprims like coffee_maker's 64-prim leaf


def test_builder_is_deterministic_and_rejects_bad_input(nnbvh_lib):
    verts, prims = ss.random_soup(500, 0, 9)
    a, b = build_tree(prims, verts), build_tree(prims, verts)
    assert a.nodes.tobytes() == b.nodes.tobytes() and a.ordered_prims.tobytes() == b.ordered_prims.tobytes()
    bad = prims.copy()
    bad["v"][3, 1] = len(verts) + 5
    with pytest.raises(NNBVHError, match="vertex index"):
        build_tree(bad, verts)
    with pytest.raises(NNBVHError, match="unknown"):
        build_tree(prims, verts, split_method="kdtree")
    h1, h2 = build_tree(prims, verts, 4, "hlbvh"), build_tree(prims, verts, 4, "hlbvh")
    assert h1.nodes.tobytes() == h2.nodes.tobytes()
    # all primitives in one point: the reference's buildUpperSAH / emitLBVH degenerate cases
    same = make_prims(np.zeros((5, 3), np.int32) + np.arange(3, dtype=np.int32))
    t = build_tree(same, verts, 4, "hlbvh")          # identical Morton codes -> a single leaf
    assert len(t.nodes) == 1 and t.nodes["nprims"][0] == 5
    with pytest.raises(NNBVHError, match="empty"):
        build_tree(prims[:0], verts)


@pytest.mark.parametrize("bad_value", [np.inf, -np.inf, np.nan])
@pytest.mark.parametrize("method", ["sah", "hlbvh", "middle", "equal"])
def test_builder_rejects_non_finite_vertices_and_bounds(nnbvh_lib, bad_value, method):
    """A +-Inf / NaN coordinate makes the reference's bucket index int(nBuckets * Offset(c))
    (aggregates.cpp:254-258) undefined (round 1's builder wrote out of bounds there): it is
    malformed input and must be an error for every split method."""
    verts, prims = ss.random_soup(64, 0, 9)
    v = verts.copy()
    v[prims["v"][17, 1], 2] = bad_value
    with pytest.raises(NNBVHError, match="non-finite"):
        build_tree(prims, v, 4, method)
    # an unreferenced bad vertex is nobody's business
    v2 = np.concatenate([verts, np.full((1, 3), bad_value, np.float32)])
    assert build_tree(prims, v2, 4, method).nodes.tobytes() == build_tree(prims, verts, 4, method).nodes.tobytes()
    # caller-supplied bounds of host / instance primitives
    allp = np.concatenate([prims, np.zeros(1, prims.dtype)])
    allp["kind"][-1], allp["id"][-1] = 3, len(prims)
    pb = np.zeros((len(allp), 6), np.float32)
    pb[-1] = [0, 0, 0, 1, 1, bad_value]
    with pytest.raises(NNBVHError, match="non-finite"):
        build_tree(allp, verts, 4, method, prim_bounds=pb)


def test_library_exports_every_symbol_the_header_declares(nnbvh_lib):
    header = open(os.path.join(ROOT, "include", "nnbvh.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(nnbvh_[a-z_]+)\s*\(", header))
    assert len(declared) >= 16
    assert declared == set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(nnbvh_lib, sym), f"{sym} declared in include/nnbvh.h but not exported"


def test_wire_struct_sizes_match_the_reference_layout():
    # LinearBVHNode is 32 bytes (aggregates.cpp:129-137); ray and hit records are 32 bytes
    assert NODE_DTYPE.itemsize == 32 and NODE_DTYPE.fields["offset"][1] == 24
    assert NODE_DTYPE.fields["nprims"][1] == 28 and NODE_DTYPE.fields["axis"][1] == 30
    assert PRIM_DTYPE.itemsize == 24


def test_scene_create_validates_tree_and_fails_loudly_without_gpu(nnbvh_lib):
    """No compute calls here: create must reject malformed trees with a message (never crash),
    and on a machine without a HIP device it must fail — there is no CPU fallback."""
    verts, prims = ss.random_soup(64, 0, 1)
    tree = build_tree(prims, verts)

    def create(nodes, ordered=tree.ordered_prims):
        return nnbvh_lib.nnbvh_scene_create(_lib.ptr(nodes), len(nodes), _lib.ptr(ordered),
                                            len(ordered), _lib.ptr(verts), len(verts), 0)

    bad = tree.nodes.copy()
    interior = np.nonzero(bad["nprims"] == 0)[0]
    bad["offset"][interior[0]] = 0  # second child pointing backwards: a cycle
    assert not create(bad) and "secondChildOffset" in _lib.last_error()
    bad = tree.nodes.copy()
    leaf = np.nonzero(bad["nprims"] > 0)[0][0]
    bad["offset"][leaf] = len(prims)  # leaf range past the primitive table
    assert not create(bad) and "out of bounds" in _lib.last_error()
    bad = tree.nodes.copy()
    bad["axis"][interior[0]] = 7
    assert not create(bad) and "axis" in _lib.last_error()
    for value in (np.float32(np.nan), None):  # NaN, or min > max on one axis: not a Bounds3f a builder emits
        bad = tree.nodes.copy()
        bad["pmin"][interior[-1], 1] = value if value is not None else bad["pmax"][interior[-1], 1] + 1
        assert not create(bad) and "min > max" in _lib.last_error()
    badp = tree.ordered_prims.copy()
    badp["kind"][0] = 99
    assert not create(tree.nodes, badp) and "kind" in _lib.last_error()
    if nnbvh_lib.nnbvh_device_count() == 0:
        assert not create(tree.nodes)
        assert "no usable HIP device" in _lib.last_error()
    # NULL handle / NULL buffers are argument errors, not crashes
    assert nnbvh_lib.nnbvh_intersect_closest(None, None, 0, None) == 1
    assert nnbvh_lib.nnbvh_scene_bounds(None, None) == 1
    nnbvh_lib.nnbvh_scene_destroy(None)
    nnbvh_lib.nnbvh_build_destroy(None)


def test_shading_mesh_validates_indices_before_touching_a_device(nnbvh_lib):
    """nnbvh_shading_mesh_create range-checks every vertex index on the host (the interaction kernel
    gathers with them); without a GPU the valid case then fails loudly on the device step."""
    from nn_bvh_amd import NNBVHError
    from nn_bvh_amd.interaction import ShadingMesh
    verts = np.zeros((6, 3), np.float32)
    with pytest.raises(NNBVHError, match="vertex index out of range"):
        ShadingMesh(verts, np.array([[0, 1, 2], [3, 4, 9]], np.int32))
    with pytest.raises(NNBVHError, match="patch vertex index out of range"):
        ShadingMesh(verts, np.array([[0, 1, 2], [-1, 0, 0]], np.int32),
                    patch_vertices=np.array([[-1, 0, 0, 0], [0, 1, 2, 6]], np.int32))
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(NNBVHError):
            ShadingMesh(verts, np.array([[0, 1, 2]], np.int32))


def test_gpu_build_entry_points_fail_loudly_without_a_gpu(nnbvh_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nn_bvh_amd import BVHAggregate, NNBVHError, build_tree_gpu, make_prims
    verts = np.random.default_rng(0).random((30, 3)).astype(np.float32)
    prims = make_prims(np.arange(30, dtype=np.int32).reshape(10, 3))
    for method in ("sah", "hlbvh"):
        with pytest.raises(NNBVHError):
            build_tree_gpu(prims, verts, split_method=method)
        with pytest.raises(NNBVHError):
            BVHAggregate.build_on_device(prims, verts, split_method=method)
    with pytest.raises(NNBVHError, match="sah.*hlbvh"):
        build_tree_gpu(prims, verts, split_method="middle")
