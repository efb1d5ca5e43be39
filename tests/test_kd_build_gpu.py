"""KdTreeAggregate construction (cpu/aggregates.cpp:798-971): what the order of EQUAL (t, type) edges can and
cannot change (CPU: the host builder with libstdc++'s std::sort against the same builder with
std::stable_sort), and the device builder against the host one byte for byte (GPU)."""
import os

import numpy as np
import pytest

import scenes_small as ss
from nn_bvh_amd import scene
from nn_bvh_amd.kdtree import build_kd_tree


def leaves_of(tree):
    """[(node index, sorted primitive ids)] of every leaf, in node order."""
    nodes, idx = tree.nodes, tree.prim_indices
    out = []
    for i in np.nonzero((nodes["flags"] & 3) == 3)[0]:
        n = int(nodes["flags"][i] >> 2)
        v = int(nodes["split_or_index"][i])
        out.append((int(i), sorted([v] if n == 1 else idx[v:v + n].tolist() if n else [])))
    return out


def scenes():
    yield "soup", ss.random_soup(3000, 0, 5)
    yield "soup+patches", ss.random_soup(1500, 300, 6)
    v, p = ss.grid_mesh(40, 7, bump=0.3)          # a connected mesh: shared vertices = many equal edges
    yield "mesh", (v, p)
    v, p = ss.coincident_centroids(120, 8)        # many identical bounds: leaves the depth limit closes
    yield "coincident", (v, p)
    vs = np.round(ss.random_soup(2500, 0, 9, extent=4.0, size=0.8)[0] * 4) / 4   # snapped to a grid
    yield "snapped", (vs.astype(np.float32), ss.random_soup(2500, 0, 9, extent=4.0, size=0.8)[1])


@pytest.mark.parametrize("max_prims", [1, 4])
def test_tie_order_changes_nothing_but_the_order_inside_leaves(max_prims):
    """aggregates.cpp:899-903 sorts with std::sort: the order among equal edges is the standard library's.
    Whatever it is, the node array (split planes, child links, leaf sizes, primitiveIndices offsets) and the
    SET of primitives of every leaf are the same."""
    some_differ = False
    for name, (verts, prims) in scenes():
        a = build_kd_tree(prims, verts, max_prims=max_prims)
        b = build_kd_tree(prims, verts, max_prims=max_prims, where="host_stable")
        leaf = (a.nodes["flags"] & 3) == 3
        multi = leaf & ((a.nodes["flags"] >> 2) > 1)
        assert a.nodes["flags"].tobytes() == b.nodes["flags"].tobytes(), name
        # split planes: equal as floats (a tie between a -0 and a +0 edge may hand either zero to the node);
        # multi-primitive leaves: the same primitiveIndices offsets; one-primitive leaves: compared as sets below
        assert np.array_equal(a.nodes["split_or_index"][~leaf].view(np.float32), b.nodes["split_or_index"][~leaf].view(np.float32)), name
        assert a.nodes["split_or_index"][multi].tobytes() == b.nodes["split_or_index"][multi].tobytes(), name
        assert len(a.prim_indices) == len(b.prim_indices) and a.depth == b.depth
        assert leaves_of(a) == leaves_of(b), name
        some_differ |= a.prim_indices.tobytes() != b.prim_indices.tobytes()
    assert some_differ or max_prims == 1  # the two orders do differ somewhere (else this test shows nothing)


@pytest.mark.gpu
@pytest.mark.parametrize("max_prims,max_depth", [(1, -1), (4, -1), (1, 6), (2, 3)])
def test_device_kd_build_equals_the_host_builder(max_prims, max_depth):
    for name, (verts, prims) in scenes():
        h = build_kd_tree(prims, verts, max_prims=max_prims, max_depth=max_depth, where="host_stable")
        g = build_kd_tree(prims, verts, max_prims=max_prims, max_depth=max_depth, where="gpu")
        assert g.nodes.tobytes() == h.nodes.tobytes(), f"{name}: node arrays differ"
        assert g.prim_indices.tobytes() == h.prim_indices.tobytes(), f"{name}: primitiveIndices differ"
        assert g.depth == h.depth and np.array_equal(g.bounds, h.bounds)


@pytest.mark.gpu
def test_device_kd_build_other_costs_and_host_primitives():
    verts, prims = ss.random_soup(2000, 100, 21)
    pb = np.zeros((len(prims), 6), np.float32)
    prims = prims.copy()
    host = np.arange(0, len(prims), 37)
    for i in host:   # a few host-only primitives with caller bounds
        c = np.random.default_rng(int(i)).uniform(-3, 3, 3).astype(np.float32)
        pb[i] = np.concatenate([c - 0.3, c + 0.3])
        prims["kind"][i] = 3
    for kw in (dict(isect_cost=80, traversal_cost=1, empty_bonus=0.2), dict(isect_cost=1, traversal_cost=4, empty_bonus=0.0),
               dict(max_prims=8)):
        h = build_kd_tree(prims, verts, prim_bounds=pb, where="host_stable", **kw)
        g = build_kd_tree(prims, verts, prim_bounds=pb, where="gpu", **kw)
        assert g.nodes.tobytes() == h.nodes.tobytes() and g.prim_indices.tobytes() == h.prim_indices.tobytes(), kw


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["killeroos", "bathroom"])
def test_device_kd_build_on_scene_blobs(name):
    if not os.path.exists(scene.blob_path(name)):
        pytest.skip(f"data/{name}.npz not present")
    verts, tris = scene.load_blob(name)
    prims = ss.make_prims(tris)
    h = build_kd_tree(prims, verts, where="host_stable")
    g = build_kd_tree(prims, verts, where="gpu")
    assert g.nodes.tobytes() == h.nodes.tobytes() and g.prim_indices.tobytes() == h.prim_indices.tobytes()
    print(f"{name}: {len(g.nodes)} nodes, {len(g.prim_indices)} indices, depth {g.depth}; device build "
          f"{g.build_ms[0]:.1f} ms (+ download {g.build_ms[1] - g.build_ms[0]:.1f} ms)")
