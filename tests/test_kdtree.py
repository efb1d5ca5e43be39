"""KdTreeAggregate (/root/reference/src/pbrt/cpu/aggregates.cpp:746-1161): the host builder's
invariants, the oracle's traversal against a tree-free brute force, the C ABI's validation (CPU) and
the device kernels against the oracle, bit for bit incl. kdNodesVisited / nTriTests (GPU)."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import NNBVHError, _lib, scene
from nn_bvh_amd.kdtree import KdTreeAggregate, build_kd_tree, kd_from_planes, prim_bounds_of


def walk(tree, n_prims):
    """Check the reference's layout and return (leaf prim lists, depth)."""
    nodes, idx = tree.nodes, tree.prim_indices
    leaves, stack, depth = [], [(0, len(nodes), 0)], 0
    seen = 0
    while stack:
        i, end, d = stack.pop()
        seen += 1
        depth = max(depth, d)
        fl = int(nodes["flags"][i])
        if fl & 3 == 3:
            assert i + 1 == end
            n = fl >> 2
            v = int(nodes["split_or_index"][i].astype(np.int32)) if n else 0
            ids = [] if n == 0 else ([v] if n == 1 else list(idx[v:v + n]))
            assert all(0 <= k < n_prims for k in ids)
            leaves.append(ids)
        else:
            above = fl >> 2
            assert i + 1 < above < end and (fl & 3) in (0, 1, 2)
            stack.append((above, end, d + 1))
            stack.append((i + 1, above, d + 1))
    assert seen == len(nodes)
    return leaves, depth


@pytest.mark.parametrize("max_prims", [1, 4])
def test_builder_layout_and_coverage(max_prims):
    verts, prims = ss.random_soup(1500, 200, 5)
    t = build_kd_tree(prims, verts, max_prims=max_prims)
    leaves, depth = walk(t, len(prims))
    assert depth == t.depth <= round(8 + 1.3 * int(np.log2(len(prims))))
    covered = set(k for ids in leaves for k in ids)
    assert covered == set(range(len(prims)))  # every primitive reaches at least one leaf
    lo, hi = prim_bounds_of(prims, verts)
    assert np.array_equal(t.bounds, np.concatenate([lo.min(0), hi.max(0)]))
    # deterministic
    t2 = build_kd_tree(prims, verts, max_prims=max_prims)
    assert t.nodes.tobytes() == t2.nodes.tobytes() and t.prim_indices.tobytes() == t2.prim_indices.tobytes()


def test_builder_known_answer_two_separated_triangles():
    # two triangles far apart along x: one split between them (empty-bonus free), two one-prim leaves
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [10, 0, 0], [11, 0, 0], [10, 1, 0]], np.float32)
    prims = ss.make_prims(np.array([[0, 1, 2], [3, 4, 5]], np.int32))
    t = build_kd_tree(prims, verts)
    assert len(t.nodes) == 3 and len(t.prim_indices) == 0
    assert int(t.nodes["flags"][0]) == (0 | (2 << 2))            # axis x, above child = node 2
    assert t.nodes["split_or_index"][:1].view(np.float32)[0] in (1.0, 10.0)
    assert [int(f) for f in t.nodes["flags"][1:]] == [3 | (1 << 2)] * 2
    assert [int(v) for v in t.nodes["split_or_index"][1:]] == [0, 1]


def test_oracle_root_interval_matches_a_literal_restatement():
    rng = np.random.default_rng(3)
    n = 4000
    b = np.sort(rng.uniform(-5, 5, (n, 2, 3)).astype(np.float32), axis=1).reshape(n, 6)
    o = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    centre = (b[:, :3] + b[:, 3:]) / 2 + rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = (centre - o).astype(np.float32)
    d[::9, 0] = 0
    d[::13, 1] = -0.0
    tmax = np.where(rng.random(n) < 0.3, np.float32(np.inf), rng.uniform(0, 20, n).astype(np.float32)).astype(np.float32)
    hit, tt = ob.bounds_t0t1(b, o, d, tmax)
    g3 = np.float32(3 * 2.0 ** -24) / (np.float32(1) - np.float32(3 * 2.0 ** -24))
    with np.errstate(all="ignore"):
        for i in range(n):
            t0, t1, ok = np.float32(0), tmax[i], True
            for k in range(3):
                inv = np.float32(1) / d[i, k]
                tn, tf = (b[i, k] - o[i, k]) * inv, (b[i, 3 + k] - o[i, k]) * inv
                if tn > tf:
                    tn, tf = tf, tn
                tf = tf * (np.float32(1) + np.float32(2) * g3)
                t0 = tn if tn > t0 else t0
                t1 = tf if tf < t1 else t1
                if t0 > t1:
                    ok = False
                    break
            assert bool(hit[i]) == ok
            if ok:
                assert tt[i].tobytes() == np.array([t0, t1], np.float32).tobytes()
    assert 0.1 < hit.mean() < 0.9


@pytest.mark.parametrize("max_prims", [1, 4])
def test_oracle_kd_traversal_agrees_with_brute_force_and_the_bvh(max_prims):
    verts, prims = ss.random_soup(1200, 150, 7)
    t = build_kd_tree(prims, verts, max_prims=max_prims)
    rays = np.concatenate([scene.random_rays(6000, verts.min(0) - 2, verts.max(0) + 2, 8),
                           scene.random_rays(2000, verts.min(0), verts.max(0), 9, tmax=0.4)])
    rays = rays[(rays["d"] != 0).all(1)]
    h = ob.kd_closest(t.nodes, t.prim_indices, prims, verts, t.bounds, rays)
    b = ob.brute_closest(prims, verts, rays)
    assert ((h["prim"] >= 0) == (b["prim"] >= 0)).all()
    assert (h["t"].view(np.uint32) == b["t"].view(np.uint32)).all()
    assert (h["prim"] != b["prim"]).mean() < 0.01
    assert (h["prim"] >= 0).mean() > 0.2 and h["nodes_visited"].max() > 10
    occ, vis, tst = ob.kd_any_hit(t.nodes, t.prim_indices, prims, verts, t.bounds, rays)
    assert (occ == (b["prim"] >= 0)).all()
    assert (vis >= 1)[occ == 1].all() and (tst >= 1)[occ == 1].all()
    # rays that miss the bounds visit nothing (aggregates.cpp:975-977)
    far = scene.random_rays(100, verts.max(0) + 5, verts.max(0) + 9, 4, tmax=1.0)
    hf = ob.kd_closest(t.nodes, t.prim_indices, prims, verts, t.bounds, far)
    assert (hf["prim"] == -1).all() and (hf["nodes_visited"] == 0).all()


def test_nss_plane_array_import_known_answer(tmp_path):
    """The npz layout of kdTree.exportTree_structure (nss_kd_tree.py:239-240): a = points, b = level-order
    planes (axis one-hot, unused, offset - 1).  Two levels: x at 0.5, then y at 0.25 / z at 0.75."""
    verts, prims = ss.random_soup(400, 0, 3, extent=1.0, size=0.05)
    planes = np.array([[1, 0, 0, 0, 0.5], [0, 1, 0, 0, 0.25], [0, 0, 1, 0, 0.75]], np.float32)
    path = tmp_path / "tree.npz"
    np.savez_compressed(path, a=np.zeros((2, 3), np.float32), b=planes)
    with np.load(path, allow_pickle=False) as f:
        t = kd_from_planes(f["b"], prims, verts)
    leaves, depth = walk(t, len(prims))
    assert depth == 2 and len(leaves) == 4 and len(t.nodes) == 7
    lo, hi = prim_bounds_of(prims, verts)
    bmin, ext = lo.min(0), hi.max(0) - lo.min(0)
    sx = np.float32(0.5) * ext[0] + bmin[0]
    assert t.nodes["split_or_index"][:1].view(np.float32)[0] == sx and int(t.nodes["flags"][0]) & 3 == 0
    below = set(np.nonzero(lo[:, 0] < sx)[0])
    assert set(leaves[0]) | set(leaves[1]) == below
    # and it traces like any other kd-tree: same t as the brute force
    rays = scene.random_rays(3000, verts.min(0) - 1, verts.max(0) + 1, 5)
    rays = rays[(rays["d"] != 0).all(1)]
    h = ob.kd_closest(t.nodes, t.prim_indices, prims, verts, t.bounds, rays)
    b = ob.brute_closest(prims, verts, rays)
    assert (h["t"].view(np.uint32) == b["t"].view(np.uint32)).all()


def test_kd_scene_create_validates_the_tree_before_touching_a_device():
    verts, prims = ss.random_soup(300, 0, 2)
    t = build_kd_tree(prims, verts)

    def create(nodes=t.nodes, idx=t.prim_indices):
        return KdTreeAggregate.from_tree(nodes, idx, prims, verts, t.bounds)

    bad = t.nodes.copy()
    interior = np.nonzero((bad["flags"] & 3) != 3)[0]
    bad["flags"][interior[0]] = (bad["flags"][interior[0]] & 3) | (len(bad) + 5 << 2)
    with pytest.raises(NNBVHError, match="above-child"):
        create(nodes=bad)
    bad = t.nodes.copy()
    one = np.nonzero(bad["flags"] == (3 | (1 << 2)))[0][0]
    bad["split_or_index"][one] = len(prims) + 3
    with pytest.raises(NNBVHError, match="primitive index out of range"):
        create(nodes=bad)
    if len(t.prim_indices):
        idx = t.prim_indices.copy()
        idx[0] = -1
        with pytest.raises(NNBVHError, match="primitiveIndices"):
            create(idx=idx)
    with pytest.raises(NNBVHError, match="unreachable|subtree"):
        create(nodes=np.concatenate([t.nodes, t.nodes[-1:]]))
    inst = prims.copy()
    inst["kind"][0] = 2
    with pytest.raises(NNBVHError, match="unsupported primitive kind"):
        KdTreeAggregate.from_tree(t.nodes, t.prim_indices, inst, verts, t.bounds)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(NNBVHError, match="no usable HIP device"):
            create()
    v = verts.copy()
    v[prims["v"][5, 0], 1] = np.inf
    with pytest.raises(NNBVHError, match="non-finite"):
        build_kd_tree(prims, v)
    assert _lib.lib().nnbvh_kd_intersect_closest(None, None, 1, None) == 1


# ---- GPU -------------------------------------------------------------------------------------
def _gpu_parity(prims, verts, tree, rays, prim_bounds=None):
    agg = KdTreeAggregate.from_tree(tree.nodes, tree.prim_indices, prims, verts, tree.bounds)
    got = agg.Intersect(rays)
    exp = ob.kd_closest(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, rays, 4)
    assert got.tobytes() == exp.tobytes(), "kd closest-hit records differ from the oracle"
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    eo, ev, et = ob.kd_any_hit(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, rays, 4)
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    assert np.array_equal(agg.IntersectP(rays), eo)
    agg.close()
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("max_prims", [1, 4])
def test_device_kd_traversal_equals_oracle_on_soups(max_prims):
    verts, prims = ss.random_soup(6000, 800, 11)
    tree = build_kd_tree(prims, verts, max_prims=max_prims)
    rays = np.concatenate([scene.random_rays(40000, verts.min(0) - 2, verts.max(0) + 2, 12),
                           scene.random_rays(10000, verts.min(0), verts.max(0), 13, tmax=0.5),
                           ss.edge_case_rays(verts, prims, 14)])
    hits = _gpu_parity(prims, verts, tree, rays)
    assert (hits["prim"] >= 0).mean() > 0.2 and hits["nodes_visited"].max() > 20


@pytest.mark.gpu
def test_device_kd_traversal_with_alpha_tested_triangles():
    """GeometricPrimitives with a constant alpha inside a KdTreeAggregate (kinds 4 / 5): the same stochastic
    test and re-trace as in the BVH kernels, once per leaf the primitive overlaps."""
    from test_alpha import alpha_scene
    verts, prims, alpha, kinds = alpha_scene(6, 2500)
    tree = build_kd_tree(prims, verts, max_prims=2)
    rays = np.concatenate([scene.random_rays(30000, verts.min(0) - 1, verts.max(0) + 1, 21),
                           scene.random_rays(8000, verts.min(0), verts.max(0), 22, tmax=0.6)])
    hits = _gpu_parity(prims, verts, tree, rays)
    mid = (alpha > 0) & (alpha < 1) & (kinds != 0)
    assert ((hits["prim"] >= 0) & mid[np.maximum(hits["prim"], 0)]).sum() > 500


@pytest.mark.gpu
def test_device_kd_traversal_with_attribute_reading_alpha_kinds():
    """nnbvh_kd_scene_create_with_attributes: alpha-tested triangles of smooth meshes (kinds 6 / 7) and alpha-tested
    bilinear patches (kinds 8 .. 15, recursion after a rejected hit included) inside a KdTreeAggregate — once per leaf
    the primitive overlaps, like every kd primitive.  Without the arrays they read such primitives void the ray."""
    from test_alpha import alpha_patch_scene, patch_uvs
    verts, prims, normals, alpha, kinds = alpha_patch_scene(43, 1500, 2500)
    rng = np.random.default_rng(4)
    prims = prims.copy()
    smooth = ((kinds == 4) | (kinds == 5)) & (rng.random(len(prims)) < 0.5)
    prims["kind"] = np.where(smooth, kinds + 2, kinds)
    kinds = prims["kind"].copy()
    uvs = patch_uvs(verts)
    tree = build_kd_tree(prims, verts, max_prims=2)
    rays = np.concatenate([scene.random_rays(40000, verts.min(0) - 1, verts.max(0) + 1, 21),
                           scene.random_rays(8000, verts.min(0), verts.max(0), 22, tmax=0.6)])
    agg = KdTreeAggregate.from_tree(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, normals=normals, uvs=uvs,
                                    prim_alpha=alpha)
    got = agg.Intersect(rays)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    try:
        ob.set_vertex_normals(normals)
        ob.set_vertex_uvs(uvs)
        ob.set_prim_alpha(alpha)
        exp = ob.kd_closest(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, rays, 4)
        eo, ev, et = ob.kd_any_hit(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, rays, 4)
    finally:
        ob.set_vertex_normals(None)
        ob.set_vertex_uvs(None)
        ob.set_prim_alpha(None)
    assert got.tobytes() == exp.tobytes()
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    assert np.array_equal(agg.IntersectP(rays), eo)
    hit_kind = kinds[np.maximum(exp["prim"], 0)]
    for k in (6, 7, 8, 9, 10, 11, 12, 13, 14, 15):
        assert ((hit_kind == k) & (exp["prim"] >= 0)).sum() > 50, k
    agg.close()
    # without the arrays: the host's
    plain = KdTreeAggregate.from_tree(tree.nodes, tree.prim_indices, prims, verts, tree.bounds)
    void = plain.Intersect(rays[:5000])["instance"] == -1
    assert void.sum() > 500
    plain.close()


@pytest.mark.gpu
def test_device_kd_traversal_mesh_host_prims_and_deep_stack():
    # connected mesh: shared edges / vertices, ties
    verts, prims = ss.grid_mesh(64, 3)
    tree = build_kd_tree(prims, verts)
    rays = np.concatenate([scene.random_rays(30000, verts.min(0) - 1, verts.max(0) + 1, 15),
                           ss.edge_case_rays(verts, prims, 16)])
    _gpu_parity(prims, verts, tree, rays)
    # host-only primitives void the rays that reach them
    verts, prims = ss.random_soup(2000, 0, 17)
    extra = np.zeros(15, prims.dtype)
    extra["kind"], extra["id"] = 3, len(prims) + np.arange(15)
    allp = np.concatenate([prims, extra])
    rng = np.random.default_rng(18)
    lo = rng.uniform(-8, 8, (len(allp), 3)).astype(np.float32)
    pb = np.concatenate([lo, lo + rng.uniform(0.5, 2, (len(allp), 3)).astype(np.float32)], 1)
    tree = build_kd_tree(allp, verts, prim_bounds=pb)
    hits = _gpu_parity(allp, verts, tree, scene.random_rays(20000, verts.min(0) - 2, verts.max(0) + 2, 19))
    assert (hits["instance"] == -1).any()
    # a long thin strip: deep tree, to-visit lists beyond the 8-entry LDS window (HBM spill path)
    verts, prims = ss.random_soup(4000, 0, 20, extent=0.5, size=0.4)
    verts = verts * np.array([400, 1, 1], np.float32)
    tree = build_kd_tree(prims, verts, max_prims=1, max_depth=40)
    o = np.zeros(8000, _lib.RAY_DTYPE)
    rng = np.random.default_rng(21)
    o["o"] = np.stack([np.full(8000, -250.0), rng.uniform(-.5, .5, 8000), rng.uniform(-.5, .5, 8000)], 1)
    o["d"] = np.stack([np.ones(8000), rng.uniform(-.002, .002, 8000), rng.uniform(-.002, .002, 8000)], 1)
    o["tmax"] = np.inf
    hits = _gpu_parity(prims, verts, tree, o)
    assert hits["nodes_visited"].max() > 60


@pytest.mark.gpu
def test_device_kd_traversal_on_a_scene_blob_and_an_nss_style_tree():
    import os
    if not os.path.exists(scene.blob_path("killeroos")):
        pytest.skip("data/killeroos.npz not present")
    verts, tris = scene.load_blob("killeroos")
    prims = ss.make_prims(tris)
    tree = build_kd_tree(prims, verts)
    rays = scene.camera_rays("killeroos", subsample=2)
    hits = _gpu_parity(prims, verts, tree, rays)
    assert (hits["prim"] >= 0).mean() > 0.5
    # learned-tree import: 4 levels of axis-cycling median-ish planes
    planes = np.zeros((15, 5), np.float32)
    for lvl in range(4):
        for k in range(2 ** lvl):
            planes[2 ** lvl - 1 + k, lvl % 3] = 1
            planes[2 ** lvl - 1 + k, 4] = (k + 0.5) / 2 ** lvl if lvl % 3 == 0 else 0.5
    nss = kd_from_planes(planes, prims, verts)
    _gpu_parity(prims, verts, nss, rays[::4])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bathroom", "crown"])
def test_device_kd_traversal_on_scene_blobs_sampled_per_ray_class(name):
    """bathroom's and crown's kd-trees (KdTreeAggregate::Create's defaults, built on the device; bathroom V = 57.5,
    T = 34 per primary ray): a 60 k-ray sample of each ray class — primary, diffuse bounce, shadow — against the
    oracle, as the BVH blob test does."""
    import os
    if not os.path.exists(scene.blob_path(name)):
        pytest.skip(f"data/{name}.npz not present")
    verts, tris = scene.load_blob(name)
    prims = ss.make_prims(tris)
    tree = build_kd_tree(prims, verts, where="gpu")
    agg = KdTreeAggregate.from_tree(tree.nodes, tree.prim_indices, prims, verts, tree.bounds)
    primary = scene.camera_rays(name, seed=1, sample=0)
    hits = agg.Intersect(primary)
    idx = np.random.default_rng(1).choice(len(primary), 60000, replace=False)
    exp = ob.kd_closest(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, primary[idx], 16)
    assert hits[idx].tobytes() == exp.tobytes(), f"{name} kd primary"
    bounce = scene.bounce_rays(primary, hits, verts, tris)
    bh = agg.Intersect(bounce)
    idx = np.random.default_rng(2).choice(len(bounce), 60000, replace=False)
    assert bh[idx].tobytes() == ob.kd_closest(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, bounce[idx],
                                              16).tobytes(), f"{name} kd bounce"
    lo, hi = verts.min(0), verts.max(0)
    shadow = scene.shadow_rays(primary, hits, verts, tris, lo + (hi - lo) * [0.3, 0.9, 0.3],
                               lo + (hi - lo) * [0.7, 1.0, 0.7])
    occ, vis, tst = agg.IntersectP(shadow, counts=True)
    idx = np.random.default_rng(3).choice(len(shadow), 60000, replace=False)
    eo, ev, et = ob.kd_any_hit(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, shadow[idx], 16)
    assert np.array_equal(occ[idx], eo) and np.array_equal(vis[idx], ev) and np.array_equal(tst[idx], et)
    assert np.array_equal(agg.IntersectP(shadow), occ)  # counting == non-counting, whole batch
    assert (hits["prim"] >= 0).mean() > (0.9 if name == "bathroom" else 0.7) and hits["nodes_visited"].mean() > 30
    agg.close()
