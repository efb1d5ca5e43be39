"""Two-level (instanced) scenes: TransformedPrimitive semantics of cpu/primitive.cpp:112-131.

CPU: the oracle's two-level traversal against explicitly transformed ("flattened") geometry —
same hits up to the rounding of the transform; its ray transform is pinned bit-for-bit to the
reference binary in test_oracle_vs_reference_live.py.  GPU: the INST kernels against the oracle,
bit-exact including node-visit counts across both levels."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import BVHAggregate, build_tree, instancing, make_prims, scene
from test_oracle_vs_reference_live import random_affine


def two_level_scene(seed=0, n_place=40):
    rng = np.random.default_rng(seed)
    # two object definitions (a bumpy grid and a small soup) + a few top-level triangles
    va, pa = ss.grid_mesh(12, seed)
    vb, pb = ss.random_soup(150, 30, seed + 1, extent=1.0, size=0.2)
    vt, pt = ss.random_soup(40, 0, seed + 2, extent=30.0, size=2.0)
    pb = pb.copy()
    pb["v"] += len(va)
    pb["v"][pb["kind"] == 0, 3] = 0
    pt = pt.copy()
    pt["v"][:, :3] += len(va) + len(vb)
    verts = np.concatenate([va, vb, vt]).astype(np.float32)
    M, Mi = random_affine(rng, n_place)
    M[:, :3, 3] = rng.uniform(-25, 25, size=(n_place, 3))
    M[:, :3, :3] *= (0.3 / np.abs(M[:, :3, :3]).max((1, 2)))[:, None, None] * rng.uniform(1, 6, (n_place, 1, 1))
    M64 = M.astype(np.float64)
    Mi = np.linalg.inv(M64).astype(np.float32)
    placements = [(int(rng.integers(0, 2)), M[j, :3].reshape(12), Mi[j, :3].reshape(12)) for j in range(n_place)]
    nodes, prims, instances, n_top = instancing.assemble_two_level(pt, verts, [pa, pb], placements)
    return verts, nodes, prims, instances, n_top, (pa, pb, pt, M64, placements)


def test_oracle_two_level_matches_flattened_geometry(nnbvh_lib):
    verts, nodes, prims, instances, n_top, (pa, pb, pt, M64, placements) = two_level_scene(3, 25)
    lo = np.array([-30, -30, -30.0])
    rays = scene.random_rays(4000, lo, -lo, 9)
    rays = rays[(rays["d"] != 0).all(1)]
    h = ob.closest_inst(nodes, prims, verts, instances, rays, 4)
    assert (h["instance"] > 0).sum() > 50 and ((h["prim"] >= 0) & (h["instance"] == 0)).sum() > 20
    # flatten: transform every placed object's vertices to render space (float64) and brute-force
    fverts, ftris, tag = [verts.astype(np.float64)], [], []
    for j, (k, m, _) in enumerate(placements):
        obj = (pa, pb)[k]
        tri = obj[obj["kind"] == 0]["v"][:, :3]
        idx, inv = np.unique(tri, return_inverse=True)
        P = verts[idx].astype(np.float64) @ M64[j, :3, :3].T + M64[j, :3, 3]
        base = sum(len(v) for v in fverts)
        fverts.append(P)
        ftris.append(inv.reshape(-1, 3) + base)
        tag += [j + 1] * len(tri)
    top_tri = pt["v"][:, :3]
    ftris.append(top_tri)
    tag += [0] * len(top_tri)
    fv = np.concatenate(fverts).astype(np.float32)
    ft = np.concatenate(ftris).astype(np.int32)
    b = ob.brute_closest(make_prims(ft), fv, rays)
    tag = np.asarray(tag)
    # patches of object b are not in the flattened set: compare only rays whose two-level hit is a triangle
    tri_hit = (h["prim"] >= 0) & np.isin(h["prim"], np.concatenate([pa["id"], pb["id"][pb["kind"] == 0], pt["id"]]))
    agree = tri_hit & (b["prim"] >= 0)
    # the child-space t is the parameter along the (un-normalised) transformed direction = the same ray
    # parameter as in render space, up to the origin shift dt and rounding
    rel = np.abs(h["t"][agree] - b["t"][agree]) / np.maximum(b["t"][agree], 1e-6)
    assert agree.sum() > 100 and np.median(rel) < 1e-5 and (rel < 1e-3).mean() > 0.97
    assert (h["instance"][agree] == tag[b["prim"][agree]]).mean() > 0.97


def test_instance_validation_errors(nnbvh_lib):
    from nn_bvh_amd import _lib
    verts, nodes, prims, instances, n_top, _ = two_level_scene(5, 6)

    def create(nodes=nodes, prims=prims, instances=instances, n_top=n_top):
        return nnbvh_lib.nnbvh_scene_create_instanced(
            _lib.ptr(nodes), len(nodes), n_top, _lib.ptr(prims), len(prims), _lib.ptr(verts), len(verts),
            _lib.ptr(instances), len(instances), 0)

    bad = instances.copy()
    bad["root"][0] = 0  # child tree inside the top-level range
    assert not create(instances=bad) and "outside the node array" in _lib.last_error()
    badp = prims.copy()
    inst_rows = np.nonzero(badp["kind"] == 2)[0]
    badp["v"][inst_rows[0], 0] = 99
    assert not create(prims=badp) and "instance index" in _lib.last_error()
    # an instance primitive inside a child tree = nesting: rejected
    child_leaf_prim = int(nodes[instances["root"][0]:]["offset"][np.nonzero(nodes[instances["root"][0]:]["nprims"] > 0)[0][0]])
    badp = prims.copy()
    badp["kind"][child_leaf_prim] = 2
    badp["v"][child_leaf_prim, 0] = 0
    assert not create(prims=badp) and "nested" in _lib.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_gpu_two_level_parity(seed):
    from test_gpu_parity import assert_hits_equal
    verts, nodes, prims, instances, n_top, _ = two_level_scene(seed, 60)
    lo = np.array([-30, -30, -30.0])
    rays = np.concatenate([scene.random_rays(30000, lo, -lo, seed + 20),
                           scene.random_rays(5000, lo, -lo, seed + 30, tmax=np.float32(1 - 1e-4)),
                           ss.edge_case_rays(verts, prims[prims["kind"] != 2], seed, 2048)])
    agg = BVHAggregate.from_tree(nodes, prims, verts, instances=instances, n_top_nodes=n_top)
    exp = ob.closest_inst(nodes, prims, verts, instances, rays, 16)
    got = agg.Intersect(rays)
    assert (exp["instance"] > 0).sum() > 500
    assert_hits_equal(got, exp, "two-level closest")
    assert (got["instance"] == exp["instance"]).all()
    eocc, evis, etst = ob.any_hit_inst(nodes, prims, verts, instances, rays, 16)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    assert (occ == eocc).all() and (vis == evis).all() and (tst == etst).all()
    assert (agg.IntersectP(rays) == eocc).all()
    agg.close()


def host_prim_scene(seed=0):
    """A soup in which every 7th primitive is declared host-only (bounds only)."""
    verts, prims = ss.random_soup(1500, 0, seed)
    prims = prims.copy()
    host = np.arange(len(prims)) % 7 == 3
    tri = verts[prims["v"][:, :3]]
    bounds = np.concatenate([tri.min(1), tri.max(1)], 1).astype(np.float32)
    prims["kind"][host] = 3
    tree = build_tree(prims, verts, prim_bounds=bounds)
    return verts, tree, host


def test_host_only_primitives_flag_rays_on_the_oracle(nnbvh_lib):
    verts, tree, host = host_prim_scene(2)
    rays = scene.random_rays(5000, verts.min(0), verts.max(0), 3)
    h = ob.closest_inst(tree.nodes, tree.ordered_prims, verts, np.zeros(0, ob.INSTANCE_DTYPE), rays)
    flagged = h["instance"] == -1
    assert 0.05 < flagged.mean() < 0.95
    # rays that never reach a host-only primitive are exactly the rays of the scene without them
    keep = tree.ordered_prims["kind"] != 3
    v2, p2 = verts, tree.ordered_prims[keep]
    t2 = build_tree(p2, v2)
    h2 = ob.closest(t2.nodes, t2.ordered_prims, v2, rays)
    ok = ~flagged & (rays["d"] != 0).all(1)
    assert (h["t"][ok].view(np.uint32) == h2["t"][ok].view(np.uint32)).all()
    assert not np.isin(h["prim"][h["prim"] >= 0], np.nonzero(host)[0]).any()  # never "hit"


@pytest.mark.gpu
def test_gpu_host_only_primitives_parity():
    from test_gpu_parity import assert_hits_equal
    verts, tree, host = host_prim_scene(4)
    rays = np.concatenate([scene.random_rays(20000, verts.min(0), verts.max(0), 5),
                           scene.random_rays(5000, verts.min(0), verts.max(0), 6, tmax=np.float32(1 - 1e-4))])
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    none = np.zeros(0, ob.INSTANCE_DTYPE)
    exp = ob.closest_inst(tree.nodes, tree.ordered_prims, verts, none, rays, 16)
    got = agg.Intersect(rays)
    assert_hits_equal(got, exp, "host prims closest")
    assert (got["instance"] == exp["instance"]).all() and (exp["instance"] == -1).mean() > 0.05
    eocc, evis, etst = ob.any_hit_inst(tree.nodes, tree.ordered_prims, verts, none, rays, 16)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    assert (occ == eocc).all() and (vis == evis).all() and (tst == etst).all()
    assert set(np.unique(eocc)) == {0, 1, 2}
    assert (agg.IntersectP(rays) == eocc).all()
    agg.close()
