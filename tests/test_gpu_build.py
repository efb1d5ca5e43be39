"""HLBVH construction on the GPU (nnbvh_build_create_gpu) against the host builder's HLBVH, which
restates the reference's buildHLBVH (cpu/aggregates.cpp:389-503, 626-723) with treelets emitted in
Morton order: the flattened LinearBVHNode array and the leaf-ordered primitive table must be
BYTE-IDENTICAL (same Morton codes, same stable sort, same splits, same bounds incl. the sign of
zeros, same DFS layout)."""
import os

import numpy as np
import pytest

import scenes_small as ss
from nn_bvh_amd import NNBVHError, build_tree, build_tree_gpu, make_prims, scene

pytestmark = pytest.mark.gpu


def same_tree(prims, verts, max_prims=4, prim_bounds=None, what="", method="hlbvh"):
    host = build_tree(prims, verts, max_prims, method, prim_bounds=prim_bounds)
    dev = build_tree_gpu(prims, verts, max_prims, prim_bounds=prim_bounds, split_method=method)
    what = f"{what} [{method}]"
    assert len(dev.nodes) == len(host.nodes), f"{what}: {len(dev.nodes)} vs {len(host.nodes)} nodes"
    assert dev.ordered_prims.tobytes() == host.ordered_prims.tobytes(), f"{what}: ordered prims differ"
    if dev.nodes.tobytes() != host.nodes.tobytes():
        a = dev.nodes.view(np.uint8).reshape(-1, 32)
        b = host.nodes.view(np.uint8).reshape(-1, 32)
        bad = np.nonzero((a != b).any(1))[0]
        raise AssertionError(f"{what}: {len(bad)} nodes differ, first {bad[:5]}: "
                             f"{dev.nodes[bad[:3]]} vs {host.nodes[bad[:3]]}")
    assert dev.depth == host.depth, f"{what}: depth {dev.depth} vs {host.depth}"
    return host


@pytest.mark.parametrize("max_prims", [1, 2, 4, 16, 255])
def test_soup_with_patches(max_prims):
    verts, prims = ss.random_soup(6000, 1500, 3)
    same_tree(prims, verts, max_prims, what=f"soup maxprims {max_prims}")


def test_connected_mesh_and_duplicate_codes():
    verts, prims = ss.grid_mesh(96, 2)
    same_tree(prims, verts, what="grid")
    # many primitives with one centroid -> one Morton code -> one big leaf whatever maxPrims is
    verts, prims = ss.coincident_centroids(400, 5)
    host = same_tree(prims, verts, what="coincident")
    assert host.nodes["nprims"].max() >= 300
    # clustered: few treelets, long runs of equal codes next to distinct ones
    rng = np.random.default_rng(9)
    v1, p1 = ss.random_soup(3000, 0, 6, extent=0.001, size=0.0004)
    v2, p2 = ss.random_soup(2000, 0, 7, extent=50.0)
    p2 = p2.copy()
    p2["v"][:, :3] += len(v1)
    p2["id"] += len(p1)
    same_tree(np.concatenate([p1, p2]), np.concatenate([v1, v2]), what="clustered")
    del rng


def test_signed_zeros_in_a_large_leaf_fold_like_the_sequential_builder():
    """700 triangles with one centroid (one Morton code, one leaf) whose bounds mix +0 and -0: the
    sequential fold keeps the first of equal values, and so must the wavefront-wide reduction."""
    rng = np.random.default_rng(12)
    n = 700
    z = np.where(rng.random((n, 3)) < 0.5, 0.0, -0.0).astype(np.float32)
    v = np.zeros((n, 3, 3), np.float32)
    v[:, 0] = z                     # the minimum corner: +-0 per axis
    v[:, 1] = [1, 1, 0]
    v[:, 2] = [1, 0, 1]
    v[:, 1, 2] = z[:, 2]
    v[:, 2, 1] = z[:, 1]
    verts = v.reshape(-1, 3)
    prims = make_prims(np.arange(3 * n, dtype=np.int32).reshape(n, 3))
    host = same_tree(prims, verts, what="signed zeros")
    assert len(host.nodes) == 1 and host.nodes["nprims"][0] == n
    assert np.signbit(verts[:, 0]).any() and not np.signbit(verts[:, 0]).all()
    # ... and next to ordinary geometry, so that the leaf hangs inside a treelet
    v2, p2 = ss.random_soup(4000, 0, 13)
    p2 = p2.copy()
    p2["v"][:, :3] += len(verts)
    p2["id"] += n
    same_tree(np.concatenate([prims, p2]), np.concatenate([verts, v2]), what="signed zeros + soup")


def test_tiny_inputs():
    for n in (1, 2, 3, 5):
        verts, prims = ss.random_soup(n, 0, 20 + n)
        same_tree(prims, verts, what=f"{n} prims")
    # two primitives in the same place: a single distinct code
    verts, prims = ss.random_soup(1, 0, 30)
    verts2 = np.concatenate([verts, verts])
    prims2 = np.concatenate([prims, prims])
    prims2["v"][1, :3] += len(verts)
    prims2["id"][1] = 1
    same_tree(prims2, verts2, what="two coincident prims")


def test_caller_supplied_bounds_for_instance_and_host_primitives():
    verts, prims = ss.random_soup(3000, 200, 8)
    rng = np.random.default_rng(4)
    extra = np.zeros(40, prims.dtype)
    extra["kind"] = np.where(np.arange(40) % 2 == 0, 2, 3)
    extra["id"] = len(prims) + np.arange(40)
    extra["v"][:, 0] = np.arange(40)
    allp = np.concatenate([prims, extra])
    lo = rng.uniform(-8, 8, (len(allp), 3)).astype(np.float32)
    pb = np.concatenate([lo, lo + rng.uniform(0.1, 2, (len(allp), 3)).astype(np.float32)], 1)
    same_tree(allp, verts, prim_bounds=pb, what="with instance/host prims")
    with pytest.raises(NNBVHError, match="need prim_bounds"):
        build_tree_gpu(allp, verts)


def test_bad_vertex_index_is_an_error_not_a_fault():
    verts, prims = ss.random_soup(500, 0, 9)
    prims = prims.copy()
    prims["v"][123, 1] = len(verts) + 7
    with pytest.raises(NNBVHError, match="vertex index out of range"):
        build_tree_gpu(prims, verts)


@pytest.mark.parametrize("method", ["sah", "hlbvh"])
@pytest.mark.parametrize("bad_value", [np.inf, np.nan])
def test_non_finite_vertex_is_an_error_on_the_device_too(method, bad_value):
    verts, prims = ss.random_soup(3000, 200, 9)
    v = verts.copy()
    v[prims["v"][1234, 0], 1] = bad_value
    with pytest.raises(NNBVHError, match="non-finite"):
        build_tree_gpu(prims, v, split_method=method)
    with pytest.raises(NNBVHError, match="non-finite"):
        build_tree(prims, v, 4, method)
    from nn_bvh_amd import BVHAggregate
    with pytest.raises(NNBVHError, match="non-finite"):
        BVHAggregate.build_on_device(prims, v, 4, method)


def test_thousands_of_single_leaf_treelets():
    """The shape of the HLBVH bring-up abort recorded in DESIGN.md (7 500 primitives, every
    primitive its own Morton code, > 3 000 treelets of one leaf each): a sparse soup spread over the
    whole Morton grid."""
    verts, prims = ss.random_soup(7500, 0, 41, extent=400.0, size=0.01)
    host = same_tree(prims, verts, what="sparse soup", method="hlbvh")
    assert (host.nodes["nprims"] > 0).sum() > 3000
    same_tree(prims, verts, what="sparse soup", method="sah")


@pytest.mark.parametrize("name", ["killeroos", "bathroom", "crown"])
def test_scene_blobs(name):
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "data", name + ".npz")):
        pytest.skip(f"data/{name}.npz not present")
    verts, tris = scene.load_blob(name)
    host = same_tree(make_prims(tris), verts, what=name)
    assert len(host.nodes) > 1000 and host.nodes["nprims"].sum() == len(tris)


# ---- SAH on the device: same tree, same leaf order as the host builder (= the reference's) -------------
@pytest.mark.parametrize("max_prims", [1, 4, 255])
def test_sah_soup_with_patches(max_prims):
    verts, prims = ss.random_soup(6000, 1500, 3)
    same_tree(prims, verts, max_prims, what=f"soup maxprims {max_prims}", method="sah")


def test_sah_small_and_degenerate_inputs():
    for n in (1, 2, 3, 5, 17, 64, 65, 1024, 1025, 2049):
        verts, prims = ss.random_soup(n, 0, 40 + n)
        same_tree(prims, verts, what=f"{n} prims", method="sah")
    verts, prims = ss.coincident_centroids(400, 5)      # one leaf: centroid bounds degenerate
    same_tree(prims, verts, what="coincident", method="sah")
    verts, prims = ss.coincident_centroids(3000, 6)     # ... above the wavefront-subtree threshold
    same_tree(prims, verts, what="coincident big", method="sah")
    verts, prims = ss.grid_mesh(96, 2)                  # many equal centroids / ties
    same_tree(prims, verts, what="grid", method="sah")


def test_sah_clustered_and_caller_bounds():
    v1, p1 = ss.random_soup(3000, 0, 6, extent=0.001, size=0.0004)
    v2, p2 = ss.random_soup(5000, 0, 7, extent=50.0)
    p2 = p2.copy()
    p2["v"][:, :3] += len(v1)
    p2["id"] += len(p1)
    same_tree(np.concatenate([p1, p2]), np.concatenate([v1, v2]), what="clustered", method="sah")
    verts, prims = ss.random_soup(3000, 200, 8)
    rng = np.random.default_rng(4)
    extra = np.zeros(40, prims.dtype)
    extra["kind"] = np.where(np.arange(40) % 2 == 0, 2, 3)
    extra["id"] = len(prims) + np.arange(40)
    allp = np.concatenate([prims, extra])
    lo = rng.uniform(-8, 8, (len(allp), 3)).astype(np.float32)
    pb = np.concatenate([lo, lo + rng.uniform(0.1, 2, (len(allp), 3)).astype(np.float32)], 1)
    same_tree(allp, verts, prim_bounds=pb, what="with instance/host prims", method="sah")


@pytest.mark.parametrize("name", ["killeroos", "coffee_maker", "bathroom", "crown"])
def test_sah_scene_blobs(name):
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "data", name + ".npz")):
        pytest.skip(f"data/{name}.npz not present")
    verts, tris = scene.load_blob(name)
    host = same_tree(make_prims(tris), verts, what=name, method="sah")
    assert host.nodes["nprims"].sum() == len(tris)


# ---- build + bake on the device: the tree never visits the host -----------------------------------------
def traces_identically(prims, verts, rays, method, prim_bounds=None, what=""):
    from nn_bvh_amd import BVHAggregate
    tree = build_tree(prims, verts, 4, method, prim_bounds=prim_bounds)
    host = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    dev = BVHAggregate.build_on_device(prims, verts, 4, method, prim_bounds=prim_bounds)
    for key in ("interior_records", "prim_slots", "depth", "device_bytes"):
        assert dev.info[key] == host.info[key], f"{what}: {key} {dev.info[key]} vs {host.info[key]}"
    assert np.array_equal(np.concatenate(dev.Bounds()), np.concatenate(host.Bounds()))
    a, b = host.Intersect(rays), dev.Intersect(rays)
    assert a.tobytes() == b.tobytes(), f"{what}: closest-hit records differ"
    for x, y in zip(host.IntersectP(rays, counts=True), dev.IntersectP(rays, counts=True)):
        assert np.array_equal(x, y), f"{what}: any-hit results differ"
    # ... and the checker is the oracle, not the product: the device-built scene against the CPU
    # restatement of BVHAggregate::Intersect / IntersectP walking the host builder's tree
    import oracle_binding as ob
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
    assert b.tobytes() == exp.tobytes(), f"{what}: device-built scene differs from the oracle (closest)"
    for x, y in zip(ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays), dev.IntersectP(rays, counts=True)):
        assert np.array_equal(x, y), f"{what}: device-built scene differs from the oracle (any hit)"
    host.close()
    dev.close()
    return a


@pytest.mark.parametrize("method", ["sah", "hlbvh"])
def test_device_built_and_baked_scene_traces_identically(method):
    verts, prims = ss.random_soup(5000, 800, 21)
    rays = np.concatenate([scene.random_rays(20000, verts.min(0) - 2, verts.max(0) + 2, 22),
                           ss.edge_case_rays(verts, prims, 23)])
    hits = traces_identically(prims, verts, rays, method, what=f"soup {method}")
    assert (hits["prim"] >= 0).mean() > 0.2
    # degenerate triangles (flag computed at bake time) and a single-leaf tree
    v2, p2 = ss.random_soup(300, 0, 24)
    v2 = v2.copy()
    v2[3:6] = v2[3]            # triangle 1 collapses to a point
    v2[8] = v2[7]              # triangle 2 to a segment
    traces_identically(p2, v2, scene.random_rays(4000, v2.min(0) - 1, v2.max(0) + 1, 25), method, what="degenerate")
    v1, p1 = ss.random_soup(1, 0, 26)
    traces_identically(p1, v1, scene.random_rays(500, v1.min(0) - 1, v1.max(0) + 1, 27), method, what="one prim")


def test_device_built_scene_with_host_only_primitives():
    verts, prims = ss.random_soup(3000, 100, 31)
    rng = np.random.default_rng(5)
    extra = np.zeros(25, prims.dtype)
    extra["kind"] = 3
    extra["id"] = len(prims) + np.arange(25)
    allp = np.concatenate([prims, extra])
    lo = rng.uniform(-8, 8, (len(allp), 3)).astype(np.float32)
    pb = np.concatenate([lo, lo + rng.uniform(0.5, 2, (len(allp), 3)).astype(np.float32)], 1)
    rays = scene.random_rays(15000, verts.min(0) - 2, verts.max(0) + 2, 32)
    hits = traces_identically(allp, verts, rays, "sah", prim_bounds=pb, what="host prims")
    assert (hits["instance"] == -1).any()
    # instance primitives are not baked on the device
    from nn_bvh_amd import BVHAggregate
    allp2 = allp.copy()
    allp2["kind"][-1] = 2
    with pytest.raises(NNBVHError, match="instance"):
        BVHAggregate.build_on_device(allp2, verts, prim_bounds=pb)


def test_device_built_crown_traces_identically():
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "data", "crown.npz")):
        pytest.skip("data/crown.npz not present")
    verts, tris = scene.load_blob("crown")
    rays = scene.camera_rays("crown", subsample=4)
    hits = traces_identically(make_prims(tris), verts, rays, "sah", what="crown")
    assert abs(hits["nodes_visited"].mean() - 99.4) < 1.0
