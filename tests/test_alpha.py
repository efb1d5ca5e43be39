"""Alpha-tested GeometricPrimitive with a constant alpha (cpu/primitive.cpp:50-84) as primitive kinds
NNBVH_PRIM_ALPHA_TRIANGLE / _FLIPPED: the oracle's restatement (HashFloat, SpawnRay and the interaction
it offsets from are each pinned to the compiled reference) and the device kernels against it."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import build_tree, scene


def alpha_scene(seed=4, n=3000):
    verts, prims = ss.random_soup(n, 0, seed, extent=6.0, size=1.2)
    rng = np.random.default_rng(seed + 1)
    prims = prims.copy()
    alpha = rng.choice(np.array([0.0, 0.25, 0.5, 0.9, 1.0, 1.5, -0.5], np.float32), len(prims))
    kinds = rng.choice(np.array([0, 4, 5], np.int32), len(prims), p=[0.3, 0.4, 0.3])
    prims["kind"] = kinds
    prims["v"][:, 3] = np.where(kinds == 0, 0, alpha.view(np.int32))
    return verts, prims, alpha, kinds


def test_oracle_alpha_semantics():
    verts, prims, alpha, kinds = alpha_scene()
    tree = build_tree(prims, verts)
    rays = scene.random_rays(20000, verts.min(0) - 1, verts.max(0) + 1, 7)
    h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
    # alpha >= 1 everywhere = plain triangles
    plain = prims.copy()
    plain["kind"] = 0
    plain["v"][:, 3] = 0
    tp = build_tree(plain, verts)
    assert tp.nodes.tobytes() == tree.nodes.tobytes()  # kinds 4 / 5 build like triangles
    hp = ob.closest(tp.nodes, tp.ordered_prims, verts, rays)
    opaque = prims.copy()
    opaque["v"][:, 3] = np.where(kinds == 0, 0, np.float32(1.0).view(np.int32))
    to = build_tree(opaque, verts)
    assert ob.closest(to.nodes, to.ordered_prims, verts, rays).tobytes() == hp.tobytes()
    # alpha <= 0 primitives are never hit; fully transparent set = scene without them (up to the extra re-tests)
    clear = prims.copy()
    clear["v"][:, 3] = np.where(kinds == 0, 0, np.float32(0.0).view(np.int32))
    tc = build_tree(clear, verts)
    hc = ob.closest(tc.nodes, tc.ordered_prims, verts, rays)
    assert (kinds[np.maximum(hc["prim"], 0)][hc["prim"] >= 0] == 0).all()
    only = build_tree(plain[kinds == 0], verts)
    ho = ob.closest(only.nodes, only.ordered_prims, verts, rays)
    assert np.array_equal(hc["prim"], ho["prim"]) and np.array_equal(hc["t"].view(np.uint32), ho["t"].view(np.uint32))
    # the stochastic cases: some hits on 0 < alpha < 1 primitives survive, some do not
    mid = (alpha > 0) & (alpha < 1) & (kinds != 0)
    hit_mid = (h["prim"] >= 0) & mid[np.maximum(h["prim"], 0)]
    was_mid = (hp["prim"] >= 0) & mid[np.maximum(hp["prim"], 0)]
    assert 0.2 * was_mid.sum() < hit_mid.sum() < was_mid.sum()
    # any-hit follows Intersect(...).has_value()
    occ, _, _ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays)
    assert np.array_equal(occ == 1, h["prim"] >= 0)


def test_oracle_alpha_retrace_is_unrolled_once_and_degenerate_rays_terminate():
    """primitive.cpp:63-69 recurses on the ray spawned off the rejected hit.  A planar triangle cannot be
    hit by it again; rays with NaN / zero components can, and the reference then recurses without bound.
    The restatement unrolls the recursion once and follows the library's contract for a second hit (the
    record is void: instance = -1 / occluded = 2) instead of overflowing the stack."""
    verts, prims, alpha, kinds = alpha_scene(9, 800)
    tree = build_tree(prims, verts)
    rng = np.random.default_rng(3)
    rays = scene.random_rays(6000, verts.min(0) - 1, verts.max(0) + 1, 11)
    special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-30], np.float32)
    for f in ("o", "d"):
        v = rays[f].copy()
        m = rng.random(v.shape) < 0.15
        v[m] = rng.choice(special, int(m.sum()))
        rays[f] = v
    h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, 4)  # must return
    occ, _, _ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, 4)
    void = h["instance"] == -1
    assert np.array_equal(occ[void & (h["prim"] < 0)] == 2, np.ones(int((void & (h["prim"] < 0)).sum()), bool))
    assert set(np.unique(h["instance"])) <= {0, -1}


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["host_tree", "device_build"])
def test_device_alpha_equals_oracle(how):
    from nn_bvh_amd import BVHAggregate
    verts, prims, alpha, kinds = alpha_scene(9, 5000)
    tree = build_tree(prims, verts)
    rays = np.concatenate([scene.random_rays(60000, verts.min(0) - 1, verts.max(0) + 1, 8),
                           scene.random_rays(10000, verts.min(0), verts.max(0), 10, tmax=0.6)])
    if how == "host_tree":
        agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    else:
        agg = BVHAggregate.build_on_device(prims, verts, 4, "sah")
    got = agg.Intersect(rays)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, 4)
    void = got["instance"] == -1   # an alpha re-trace that hit: never for planar triangles
    assert void.sum() == 0
    assert got.tobytes() == exp.tobytes()
    assert (kinds[np.maximum(exp["prim"], 0)][exp["prim"] >= 0] != 0).mean() > 0.3
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    eo, ev, et = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, 4)
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    assert np.array_equal(agg.IntersectP(rays), eo)
    agg.close()


@pytest.mark.gpu
def test_device_alpha_inside_instances():
    """the alpha test hashes the ray the primitive is given: inside an instance, the instance-space ray"""
    from nn_bvh_amd import BVHAggregate, instancing
    from test_oracle_vs_reference_live import random_affine
    verts, prims, alpha, kinds = alpha_scene(12, 800)
    rng = np.random.default_rng(3)
    n_place = 6
    M, _ = random_affine(rng, n_place)
    M[:, :3, 3] = rng.uniform(-15, 15, size=(n_place, 3))
    Mi = np.linalg.inv(M.astype(np.float64)).astype(np.float32)
    placements = [(0, M[j, :3].reshape(12), Mi[j, :3].reshape(12)) for j in range(n_place)]
    top = prims[:0]
    nodes, aprims, instances, n_top = instancing.assemble_two_level(top, verts, [prims], placements)
    agg = BVHAggregate.from_tree(nodes, aprims, verts, instances=instances, n_top_nodes=n_top)
    rays = scene.random_rays(40000, [-25, -25, -25], [25, 25, 25], 13)
    got = agg.Intersect(rays)
    exp = ob.closest_inst(nodes, aprims, verts, instances, rays, 4)
    assert (got["instance"] == -1).sum() == 0 and got.tobytes() == exp.tobytes()
    assert (got["instance"] > 0).mean() > 0.1
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    eo, ev, et = ob.any_hit_inst(nodes, aprims, verts, instances, rays, 4)
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    agg.close()


def smooth_alpha_scene(seed=21, n=5000):
    """alpha_scene with two thirds of the alpha-tested triangles on a 'smooth mesh' (kinds 6 / 7) and random
    per-vertex shading normals — half of them pointing to the other side of the triangle than its geometric
    normal, so that FaceForward(n, ns) flips the offset direction, and a few zero normals (ns falls back to n)."""
    verts, prims, alpha, kinds = alpha_scene(seed, n)
    rng = np.random.default_rng(seed + 7)
    prims = prims.copy()
    smooth = (kinds != 0) & (rng.random(len(prims)) < 0.67)
    prims["kind"] = np.where(smooth, kinds + 2, kinds)   # 4 -> 6, 5 -> 7
    normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    normals[rng.random(len(verts)) < 0.02] = 0
    return verts, prims, normals, alpha, prims["kind"].copy()


def test_oracle_smooth_alpha_offsets_along_the_face_forwarded_normal():
    verts, prims, normals, alpha, kinds = smooth_alpha_scene()
    tree = build_tree(prims, verts)
    rays = scene.random_rays(20000, verts.min(0) - 1, verts.max(0) + 1, 7)
    try:
        ob.set_vertex_normals(normals)
        h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
        # the same scene with every normal on the geometric normal's side gives the flat kinds' records ...
        flat = prims.copy()
        flat["kind"] = np.where(kinds >= 6, kinds - 2, kinds)
        tf = build_tree(flat, verts)
        assert tf.nodes.tobytes() == tree.nodes.tobytes()  # kinds 6 / 7 build like triangles
        hf = ob.closest(tf.nodes, tf.ordered_prims, verts, rays)
    finally:
        ob.set_vertex_normals(None)
    # ... and which primitive is hit never depends on the normals (the re-trace against the same planar
    # triangle misses whichever side it is offset to); only void flags could differ, and there are none
    assert (h["instance"] == -1).sum() == 0 and h.tobytes() == hf.tobytes()
    assert ((kinds[np.maximum(h["prim"], 0)] >= 6) & (h["prim"] >= 0)).sum() > 500
    # without normals the smooth kinds cannot be evaluated by the library; the oracle then treats them as flat


@pytest.mark.gpu
def test_device_smooth_alpha_equals_oracle():
    """NNBVH_PRIM_ALPHA_TRIANGLE_SMOOTH: the re-trace origin is offset along FaceForward(n, ns) (shapes.h:939-951).
    Records and counters bit-equal to the oracle; and the offset origin itself is exercised through rays that
    start exactly ON an alpha triangle's re-trace path: the re-trace origins of flat and smooth kinds differ."""
    from nn_bvh_amd import BVHAggregate
    from nn_bvh_amd._lib import NNBVHError
    verts, prims, normals, alpha, kinds = smooth_alpha_scene(23, 6000)
    tree = build_tree(prims, verts)
    rays = np.concatenate([scene.random_rays(60000, verts.min(0) - 1, verts.max(0) + 1, 8),
                           scene.random_rays(10000, verts.min(0), verts.max(0), 10, tmax=0.6)])
    with pytest.raises(NNBVHError, match="normals"):
        BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, normals=normals)
    got = agg.Intersect(rays)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    try:
        ob.set_vertex_normals(normals)
        exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, 4)
        eo, ev, et = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, 4)
    finally:
        ob.set_vertex_normals(None)
    assert got.tobytes() == exp.tobytes()
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    assert np.array_equal(agg.IntersectP(rays), eo)
    assert ((kinds[np.maximum(exp["prim"], 0)] >= 6) & (exp["prim"] >= 0)).mean() > 0.1
    dev = BVHAggregate.build_on_device(prims, verts, 4, "sah", normals=normals)
    assert dev.Intersect(rays).tobytes() == exp.tobytes()
    dev.close()
    agg.close()


# ---- alpha-tested BILINEAR PATCHES (prim kinds 8 .. 11) ---------------------------------------------------------------
def alpha_patch_scene(seed=31, n_tris=1500, n_patches=2500):
    """Triangles (plain / alpha-tested) and strongly twisted bilinear patches — a ray can cross such a patch twice, so
    the re-trace after a rejected hit does find it again — of all five kinds (1, 8 .. 11), random per-vertex normals
    (half of them on the other side than the geometric normal, a few zero), one constant alpha per primitive."""
    rng = np.random.default_rng(seed)
    verts, prims = ss.random_soup(n_tris, n_patches, seed, extent=6.0, size=1.2)
    verts = verts.copy()
    pv = prims["v"][n_tris:]
    verts[pv[:, 3]] += rng.uniform(-1.5, 1.5, size=(n_patches, 3)).astype(np.float32)  # twist the fourth corner
    prims = prims.copy()
    kinds = prims["kind"].copy()
    kinds[:n_tris] = rng.choice(np.array([0, 4, 5], np.int32), n_tris, p=[0.4, 0.3, 0.3])
    kinds[n_tris:] = rng.choice(np.array([1, 8, 9, 10, 11, 12, 13, 14, 15], np.int32), n_patches,
                                p=[0.12] + [0.11] * 8)
    alpha = rng.choice(np.array([0.0, 0.25, 0.5, 0.9, 1.0, 1.5, -0.5], np.float32), len(prims))
    prims["kind"] = kinds
    tri_alpha = (kinds == 4) | (kinds == 5)
    prims["v"][tri_alpha, 3] = alpha[tri_alpha].view(np.int32)
    normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    normals[rng.random(len(verts)) < 0.02] = 0
    return verts, prims, normals, alpha, kinds


def patch_uvs(verts, seed=3):
    """(u, v) per vertex: random, with a few coincident pairs so that the 1e-8 derivative tests of shapes.h:1423-1426
    and the cross(dpds, dpdt) == 0 fallback are reached"""
    rng = np.random.default_rng(seed)
    uv = rng.random((len(verts), 2)).astype(np.float32)
    same = rng.random(len(verts)) < 0.08
    uv[same] = np.float32(0.5)
    uv[rng.random(len(verts)) < 0.05, 0] = np.float32(0.25)
    return uv


def _oracle_alpha_patch(tree, verts, normals, alpha_ordered, rays, nthreads=4):
    try:
        ob.set_vertex_normals(normals)
        ob.set_vertex_uvs(patch_uvs(verts))
        ob.set_prim_alpha(alpha_ordered)
        h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads)
        occ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, nthreads)
    finally:
        ob.set_vertex_normals(None)
        ob.set_vertex_uvs(None)
        ob.set_prim_alpha(None)
    return h, occ


def test_oracle_alpha_patch_recursion():
    verts, prims, normals, alpha, kinds = alpha_patch_scene()
    tree = build_tree(prims, verts)
    order = tree.ordered_prims["id"]
    rays = scene.random_rays(40000, verts.min(0) - 1, verts.max(0) + 1, 7)
    h, (occ, _, _) = _oracle_alpha_patch(tree, verts, normals, alpha[order], rays)
    # kinds 8 .. 11 build like patches; with alpha >= 1 everywhere they ARE plain patches
    plain = prims.copy()
    plain["kind"] = np.where(kinds >= 8, 1, kinds)
    tp = build_tree(plain, verts)
    assert tp.nodes.tobytes() == tree.nodes.tobytes()
    opaque = np.where(kinds >= 8, np.float32(1.0), alpha)
    ho, _ = _oracle_alpha_patch(tree, verts, normals, opaque[order], rays)
    hp = ob.closest(tp.nodes, tp.ordered_prims, verts, rays, 4)
    assert ho.tobytes() == hp.tobytes()
    # alpha <= 0: the patch is crossed (every crossing costs a test and a re-trace), never hit
    clear = np.where(kinds >= 8, np.float32(0.0), alpha)
    hc, _ = _oracle_alpha_patch(tree, verts, normals, clear[order], rays)
    hit_kind = kinds[np.maximum(hc["prim"], 0)]
    assert (hit_kind[hc["prim"] >= 0] < 8).all()
    # the recursion is real: rays whose FIRST crossing of a mid-alpha patch was rejected and whose SECOND crossing of
    # the same patch was accepted report that patch at a larger t than the plain scene does
    mid = (alpha > 0) & (alpha < 1) & (kinds >= 8)
    same = (h["prim"] >= 0) & (h["prim"] == hp["prim"]) & mid[np.maximum(h["prim"], 0)]
    second = same & (h["t"] > hp["t"] * 1.0001)
    assert second.sum() > 20, second.sum()
    assert (h["prim_tests"][second] > hp["prim_tests"][second]).all()
    # no void records on ordinary rays (three re-traces are enough for a surface a line meets twice at most)
    assert (h["instance"] == -1).sum() == 0
    # any-hit follows Intersect(...).has_value()
    assert np.array_equal(occ == 1, h["prim"] >= 0)
    # the normals matter only through the offset direction: flat kinds (8 / 9) never read them
    flat_only = prims.copy()
    flat_only["kind"] = np.where(kinds >= 8, 8 + ((kinds - 8) & ~2), kinds)  # the smooth bit off
    tf = build_tree(flat_only, verts)
    hf, _ = _oracle_alpha_patch(tf, verts, None, alpha[order], rays)
    hf2, _ = _oracle_alpha_patch(tf, verts, normals, alpha[order], rays)
    assert hf.tobytes() == hf2.tobytes()


def test_oracle_alpha_patch_degenerate_rays_terminate():
    verts, prims, normals, alpha, kinds = alpha_patch_scene(33, 300, 600)
    tree = build_tree(prims, verts)
    rng = np.random.default_rng(3)
    rays = scene.random_rays(6000, verts.min(0) - 1, verts.max(0) + 1, 11)
    special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-30], np.float32)
    for f in ("o", "d"):
        v = rays[f].copy()
        m = rng.random(v.shape) < 0.15
        v[m] = rng.choice(special, int(m.sum()))
        rays[f] = v
    h, (occ, _, _) = _oracle_alpha_patch(tree, verts, normals, alpha[tree.ordered_prims["id"]], rays)  # must return
    assert set(np.unique(h["instance"])) <= {0, -1}
    void = (h["instance"] == -1) & (h["prim"] < 0)
    assert (occ[void] == 2).all()


@pytest.mark.gpu
def test_device_alpha_patches_equal_oracle():
    """NNBVH_PRIM_ALPHA_PATCH[_FLIPPED / _SMOOTH / _SMOOTH_FLIPPED]: records and counters bit-equal to the oracle,
    including the rays that cross a patch twice (re-trace accepted at the second crossing: accumulated tHit), and the
    degenerate rays (NaN / inf / zero components)."""
    from nn_bvh_amd import BVHAggregate
    from nn_bvh_amd._lib import NNBVHError
    verts, prims, normals, alpha, kinds = alpha_patch_scene(35, 3000, 5000)
    tree = build_tree(prims, verts)
    a_ord = alpha[tree.ordered_prims["id"]]
    rng = np.random.default_rng(5)
    rays = np.concatenate([scene.random_rays(80000, verts.min(0) - 1, verts.max(0) + 1, 8),
                           scene.random_rays(10000, verts.min(0), verts.max(0), 10, tmax=0.6)])
    weird = scene.random_rays(6000, verts.min(0) - 1, verts.max(0) + 1, 11)
    special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-30], np.float32)
    for f in ("o", "d"):
        v = weird[f].copy()
        m = rng.random(v.shape) < 0.15
        v[m] = rng.choice(special, int(m.sum()))
        weird[f] = v
    rays = np.concatenate([rays, weird])
    uvs = patch_uvs(verts)
    with pytest.raises(NNBVHError, match="alpha"):
        BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, normals=normals)
    with pytest.raises(NNBVHError, match="normals"):
        BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, prim_alpha=a_ord, uvs=uvs)
    with pytest.raises(NNBVHError, match="uvs"):
        BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, prim_alpha=a_ord, normals=normals)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, normals=normals, prim_alpha=a_ord, uvs=uvs)
    got = agg.Intersect(rays)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    exp, (eo, ev, et) = _oracle_alpha_patch(tree, verts, normals, a_ord, rays)
    # built on the device: the per-primitive alpha follows the primitives through the build (caller's order in)
    dev = BVHAggregate.build_on_device(prims, verts, 4, "sah", normals=normals, prim_alpha=alpha, uvs=uvs)
    assert dev.Intersect(rays).tobytes() == got.tobytes()
    assert np.array_equal(dev.IntersectP(rays), occ)
    dev.close()
    # ... and through the device HLBVH builder (another primitive order: the gather of the per-primitive alpha)
    th = build_tree(prims, verts, 4, "hlbvh")
    dev = BVHAggregate.build_on_device(prims, verts, 4, "hlbvh", normals=normals, prim_alpha=alpha, uvs=uvs)
    sub = rays[:30000]
    eh, _ = _oracle_alpha_patch(th, verts, normals, alpha[th.ordered_prims["id"]], sub)
    assert dev.Intersect(sub).tobytes() == eh.tobytes()
    dev.close()
    n_ord = len(rays) - len(weird)
    assert got[:n_ord].tobytes() == exp[:n_ord].tobytes()
    # the degenerate rays: a NaN ray can be accepted with NaN t / barycentrics (shapes.cpp:239-266); x86 and gfx950
    # differ in the sign of the default NaN, so there a NaN equals a NaN and everything else is held to the bit
    for f in ("prim", "nodes_visited", "prim_tests", "instance"):
        assert np.array_equal(got[f], exp[f]), f
    for f in ("t", "b0", "b1", "b2"):
        same = (got[f].view(np.uint32) == exp[f].view(np.uint32)) | (np.isnan(got[f]) & np.isnan(exp[f]))
        assert same.all(), f
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    assert np.array_equal(agg.IntersectP(rays), eo)
    hit_kind = kinds[np.maximum(exp["prim"], 0)]
    for k in range(8, 16):
        assert ((hit_kind == k) & (exp["prim"] >= 0)).sum() > 150, k
    # the twice-crossed patches are in the sample: more tests than the plain scene needs
    plain = prims.copy()
    plain["kind"] = np.where(kinds >= 8, 1, kinds)
    tp = build_tree(plain, verts)
    hp = ob.closest(tp.nodes, tp.ordered_prims, verts, rays, 4)
    second = (exp["prim"] >= 0) & (exp["prim"] == hp["prim"]) & (exp["t"] > hp["t"] * 1.0001) & (hit_kind >= 8)
    assert second.sum() > 20
    agg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("animated", [False, True])
def test_device_alpha_patches_and_smooth_triangles_inside_instances(animated):
    """nnbvh_scene_create_instanced_with_attributes: the alpha-tested kinds that read per-vertex normals / uvs and the
    per-primitive alpha (smooth triangles 6 / 7, patches 8 .. 15) inside TransformedPrimitive / AnimatedPrimitive
    instances — the alpha test hashes the instance-space ray, the re-trace runs in instance space."""
    from nn_bvh_amd import BVHAggregate, instancing
    from test_oracle_vs_reference_live import random_affine
    verts, prims, normals, alpha, kinds = alpha_patch_scene(41, 500, 700)
    rng = np.random.default_rng(3)
    prims = prims.copy()
    tri_alpha = (kinds == 4) | (kinds == 5)
    smooth = tri_alpha & (rng.random(len(prims)) < 0.5)
    prims["kind"] = np.where(smooth, kinds + 2, kinds)   # some of the alpha triangles on a smooth mesh
    uvs = patch_uvs(verts)
    anims = oa = None
    if animated:
        import test_animated as ta
        _, _, _, _, _, _, anims, oa, placements = ta.animated_scene(4, 24)
        nodes, aprims, instances, n_top = ta.rebuild_with_motion_bounds(verts, prims, placements, anims, oa)
    else:
        n_place = 6
        M, _ = random_affine(rng, n_place)
        M[:, :3, 3] = rng.uniform(-15, 15, size=(n_place, 3))
        Mi = np.linalg.inv(M.astype(np.float64)).astype(np.float32)
        placements = [(0, M[j, :3].reshape(12), Mi[j, :3].reshape(12)) for j in range(n_place)]
        nodes, aprims, instances, n_top = instancing.assemble_two_level(prims[:0], verts, [prims], placements)
    obj = aprims["kind"] != 2
    a_ord = np.zeros(len(aprims), np.float32)
    a_ord[obj] = alpha[aprims["id"][obj]]   # the object's primitives keep their ids = positions in `prims`
    assert np.array_equal(aprims["kind"][obj], prims["kind"][aprims["id"][obj]])
    rays = scene.random_rays(60000, [-25, -25, -25], [25, 25, 25], 13)
    if animated:
        rays["time"] = np.random.default_rng(6).uniform(-0.2, 1.2, len(rays)).astype(np.float32)
    agg = BVHAggregate.from_tree(nodes, aprims, verts, instances=instances, n_top_nodes=n_top, animated=anims,
                                 normals=normals, uvs=uvs, prim_alpha=a_ord)
    got = agg.Intersect(rays)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    try:
        ob.set_vertex_normals(normals)
        ob.set_vertex_uvs(uvs)
        ob.set_prim_alpha(a_ord)
        if animated:
            ob.set_sin_mode(1)  # the device's sine (the path's documented exception, tests/test_animated.py)
            exp = ob.closest_anim(nodes, aprims, verts, instances, oa, rays, 4)
            eo, ev, et = ob.any_hit_anim(nodes, aprims, verts, instances, oa, rays, 4)
        else:
            exp = ob.closest_inst(nodes, aprims, verts, instances, rays, 4)
            eo, ev, et = ob.any_hit_inst(nodes, aprims, verts, instances, rays, 4)
    finally:
        ob.set_sin_mode(0)
        ob.set_vertex_normals(None)
        ob.set_vertex_uvs(None)
        ob.set_prim_alpha(None)
    assert got.tobytes() == exp.tobytes()
    assert np.array_equal(occ, eo) and np.array_equal(vis, ev) and np.array_equal(tst, et)
    hit_kind = prims["kind"][np.maximum(exp["prim"], 0)]
    inside = (exp["prim"] >= 0) & (exp["instance"] > 0)
    for k in (6, 7, 8, 11, 12, 15):
        assert (inside & (hit_kind == k)).sum() > 20, k
    agg.close()
