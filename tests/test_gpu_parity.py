"""GPU parity: the HIP path (through the C ABI) against the oracle on identical seeded inputs.

Bar (BASELINE.json north_star): hit primitive ids and node-visit counts bit-exact;
t within 1e-5 relative — these tests demand more: t and barycentrics BIT-EQUAL, and
primitive-test counts equal too."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import BVHAggregate, build_tree, make_rays, scene

pytestmark = pytest.mark.gpu

T_REL_TOL = 1e-5  # the stated tolerance; asserted as an upper bound next to the bit test


def assert_hits_equal(got, exp, what=""):
    for f in ("prim", "nodes_visited", "prim_tests"):
        bad = np.nonzero(got[f] != exp[f])[0]
        assert len(bad) == 0, f"{what}: {f} differs on {len(bad)} rays, first {bad[:5]}: " \
                              f"{got[f][bad[:5]]} vs {exp[f][bad[:5]]}"
    hit = exp["prim"] >= 0
    with np.errstate(all="ignore"):
        rel = np.abs(got["t"][hit].astype(np.float64) - exp["t"][hit]) / np.abs(exp["t"][hit])
    # (a NaN ray can be accepted with a NaN t, shapes.cpp:239-266: those are held to the bit pattern below)
    assert (np.isnan(exp["t"][hit]) | (rel <= T_REL_TOL)).all(), f"{what}: t outside 1e-5 relative"
    for f in ("t", "b0", "b1", "b2"):
        gb, eb = got[f].view(np.uint32), exp[f].view(np.uint32)
        bad = np.nonzero(gb != eb)[0]
        assert len(bad) == 0, f"{what}: {f} not bit-equal on {len(bad)} rays, first {bad[:5]}"


def check_scene(verts, prims, rays, what, max_prims=4, split="sah", window=None):
    tree = build_tree(prims, verts, max_prims, split)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    if window:
        agg.set_option("stack_window", window)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    got = agg.Intersect(rays)
    assert_hits_equal(got, exp, what + " closest")
    eocc, evis, etst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    assert (occ == eocc).all(), what + " any: occluded differs"
    assert (vis == evis).all(), what + " any: nodes_visited differs"
    assert (tst == etst).all(), what + " any: prim_tests differs"
    occ2 = agg.IntersectP(rays)
    assert (occ2 == eocc).all(), what + " any (no counts): occluded differs"
    agg.close()
    return exp


@pytest.mark.parametrize("seed", [0, 1])
def test_triangle_soup(seed):
    verts, prims = ss.random_soup(3000, 0, seed)
    rays = np.concatenate([scene.random_rays(6000, verts.min(0) - 2, verts.max(0) + 2, seed),
                           ss.edge_case_rays(verts, prims, seed)])
    exp = check_scene(verts, prims, rays, f"soup{seed}")
    assert (exp["prim"] >= 0).mean() > 0.2


def test_mixed_triangles_and_patches():
    verts, prims = ss.random_soup(1500, 1500, 3)
    rays = np.concatenate([scene.random_rays(6000, verts.min(0) - 2, verts.max(0) + 2, 3),
                           ss.edge_case_rays(verts, prims, 3)])
    check_scene(verts, prims, rays, "mixed")


def test_connected_mesh_shared_edges_and_vertices():
    verts, prims = ss.grid_mesh(64, 1)
    rays = np.concatenate([ss.edge_case_rays(verts, prims, 5, 8192),
                           scene.random_rays(4000, [-6, -3, -6], [6, 3, 6], 9)])
    check_scene(verts, prims, rays, "grid")


def test_big_leaves_and_single_leaf_tree():
    verts, prims = ss.coincident_centroids(300, 2)
    rays = scene.random_rays(3000, [-1, 0, 1], [3, 4, 5], 4)
    check_scene(verts, prims, rays, "coincident")
    # one primitive: the root is a leaf
    v1, p1 = ss.random_soup(1, 0, 7)
    check_scene(v1, p1, scene.random_rays(500, v1.min(0) - 1, v1.max(0) + 1, 8), "single")
    # maxnodeprims 1 and 255, other split methods
    verts, prims = ss.random_soup(1200, 100, 11)
    rays = scene.random_rays(3000, verts.min(0), verts.max(0), 12)
    check_scene(verts, prims, rays, "maxprims1", max_prims=1)
    check_scene(verts, prims, rays, "maxprims255", max_prims=255)
    check_scene(verts, prims, rays, "middle", split="middle")
    check_scene(verts, prims, rays, "equal", split="equal")
    check_scene(verts, prims, rays, "hlbvh", split="hlbvh")


@pytest.mark.parametrize("window", [4, 8, 16])
def test_stack_window_spill_is_exact(window):
    """A deep, skewed tree (equal-count splits of a long thin strip, leaves of 1) with rays
    along the strip keeps dozens of nodes pending: window 4 must spill to HBM and still agree."""
    rng = np.random.default_rng(5)
    n = 4096
    x = np.cumsum(rng.uniform(0.01, 1.0, n)).astype(np.float32)
    c = np.stack([x, np.zeros(n, np.float32), np.zeros(n, np.float32)], 1)[:, None, :]
    verts = (c + rng.uniform(-0.5, 0.5, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    from nn_bvh_amd import make_prims
    prims = make_prims(np.arange(3 * n, dtype=np.int32).reshape(n, 3))
    o = np.stack([np.full(2000, -5.0), rng.uniform(-0.4, 0.4, 2000), rng.uniform(-0.4, 0.4, 2000)], 1)
    d = np.stack([np.ones(2000), rng.uniform(-1e-3, 1e-3, 2000), rng.uniform(-1e-3, 1e-3, 2000)], 1)
    rays = np.concatenate([make_rays(o, d), make_rays(o + [x[-1] + 10, 0, 0], -d)])
    exp = check_scene(verts, prims, rays, f"strip w{window}", max_prims=1, split="equal",
                      window=window)
    assert exp["nodes_visited"].max() > 100


def test_golden_traversal_fixture():
    path = os.path.join(os.path.dirname(__file__), "golden", "traversal_small.npz")
    g = np.load(path)
    agg = BVHAggregate.from_tree(g["nodes"], g["ordered_prims"], g["verts"])
    got = agg.Intersect(g["rays"])
    assert_hits_equal(got, g["hits"], "golden closest")
    occ, vis, tst = agg.IntersectP(g["rays"], counts=True)
    assert (occ == g["occ"]).all() and (vis == g["occ_visited"]).all() and (tst == g["occ_tests"]).all()
    agg.close()


def test_results_independent_of_tuning_and_order():
    """Scheduling knobs and ray order must never change a ray's result."""
    verts, prims = ss.random_soup(4000, 200, 21)
    rays = scene.random_rays(20000, verts.min(0), verts.max(0), 22)
    agg = BVHAggregate(prims, verts)
    base = agg.Intersect(rays)
    perm = np.random.default_rng(0).permutation(len(rays))
    shuffled = agg.Intersect(rays[perm])
    assert (shuffled.tobytes() == base[perm].tobytes())
    for key, val in (("xcd_queues", 0), ("refill_weight", 1), ("refill_weight", 64),
                     ("prim_weight", 1), ("prim_weight", 64), ("blocks_per_cu", 1),
                     ("stack_window", 4), ("int_repeat", 1), ("int_repeat", 5), ("prim_repeat", 1),
                     ("prim_repeat", 4)):
        agg.set_option(key, val)
        assert agg.Intersect(rays).tobytes() == base.tobytes(), f"{key}={val} changed results"
    agg.close()


def test_empty_and_tiny_batches():
    verts, prims = ss.random_soup(100, 0, 1)
    agg = BVHAggregate(prims, verts)
    assert len(agg.Intersect(make_rays(np.zeros((0, 3)), np.zeros((0, 3))))) == 0
    tree = build_tree(prims, verts)
    for n in (1, 63, 64, 65, 257):
        rays = scene.random_rays(n, verts.min(0), verts.max(0), n)
        exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
        assert_hits_equal(agg.Intersect(rays), exp, f"n={n}")
    lo, hi = agg.Bounds()
    assert (lo == tree.nodes["pmin"][0]).all() and (hi == tree.nodes["pmax"][0]).all()
    agg.close()


def test_degenerate_rays_nan_inf_zero_direction():
    """Zero direction, NaN/Inf components: same (miss/hit) decisions as the reference's
    comparisons give, and the kernel must terminate."""
    verts, prims = ss.grid_mesh(16, 3)
    o = np.array([[0, 5, 0]] * 8, np.float32)
    d = np.array([[0, 0, 0], [np.nan, -1, 0], [0, -np.inf, 0], [np.inf, np.inf, np.inf],
                  [0, -1, 0], [0, 1, 0], [1e-38, -1e-38, 1e-38], [-0.0, -1, -0.0]], np.float32)
    rays = make_rays(o, d)
    rays["tmax"][4] = np.nan
    check_scene(verts, prims, rays, "degenerate rays")


def test_rays_on_box_planes_with_zero_and_infinite_components():
    """The interior step carries the slab test's verdict as one float (trace_math.h slab_entry_key).  Its
    equality with the reference's test rests on how NaNs fall: origins exactly ON slab planes (vertex
    coordinates are node bounds) with zero, negative-zero and infinite direction components produce
    0 * inf in every combination; hits and both counters must still be the oracle's."""
    verts, prims = ss.grid_mesh(12, 5)
    rng = np.random.default_rng(3)
    n = 6000
    o = verts[rng.integers(0, len(verts), n)].copy()
    keep = rng.random((n, 3)) < 0.7  # the others: off the plane along that axis
    o = np.where(keep, o, o + rng.choice(np.array([-1.5, -0.25, 0.25, 2.0], np.float32), (n, 3))).astype(np.float32)
    d = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, 0.5, np.inf, -np.inf, 1e-30], np.float32), (n, 3)).astype(np.float32)
    rays = make_rays(o, d)
    rays["tmax"] = rng.choice(np.array([np.inf, 1.0, 3.0, 0.0, 1e-30], np.float32), n)
    exp = check_scene(verts, prims, rays, "rays on box planes")
    assert 0.02 < (exp["prim"] >= 0).mean() < 0.98
    check_scene(verts, prims, rays, "rays on box planes, 1-primitive leaves", max_prims=1)


@pytest.mark.parametrize("name", ["killeroos", "coffee_maker", "bathroom", "crown"])
def test_reference_scene_blobs_full_size(name):
    """Full-size scenes (when the git-ignored blob travelled): oracle parity on a sample of each
    ray class, plus size-independent properties on the full primary batch."""
    if not os.path.exists(scene.blob_path(name)):
        pytest.skip(f"data/{name}.npz not present")
    verts, tris = scene.load_blob(name)
    from nn_bvh_amd import make_prims
    prims = make_prims(tris)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    primary = scene.camera_rays(name, seed=1, sample=0)
    hits = agg.Intersect(primary)
    # sample parity (oracle on 60k rays per class keeps the CPU side in seconds)
    idx = np.random.default_rng(1).choice(len(primary), 60000, replace=False)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, primary[idx], nthreads=16)
    assert_hits_equal(hits[idx], exp, f"{name} primary")
    bounce = scene.bounce_rays(primary, hits, verts, tris)
    bh = agg.Intersect(bounce)
    idx = np.random.default_rng(2).choice(len(bounce), 60000, replace=False)
    assert_hits_equal(bh[idx], ob.closest(tree.nodes, tree.ordered_prims, verts, bounce[idx], 16),
                      f"{name} bounce")
    lo, hi = verts.min(0), verts.max(0)
    shadow = scene.shadow_rays(primary, hits, verts, tris, lo + (hi - lo) * [0.3, 0.9, 0.3],
                               lo + (hi - lo) * [0.7, 1.0, 0.7])
    occ, vis, tst = agg.IntersectP(shadow, counts=True)
    idx = np.random.default_rng(3).choice(len(shadow), 60000, replace=False)
    eocc, evis, etst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, shadow[idx], 16)
    assert (occ[idx] == eocc).all() and (vis[idx] == evis).all() and (tst[idx] == etst).all()
    # properties on the full batches
    assert ((hits["prim"] >= 0) == (agg.IntersectP(primary) == 1)).all()  # any-hit == has closest hit
    assert (occ == agg.IntersectP(shadow)).all()                          # counting == non-counting
    again = primary.copy()
    m = hits["prim"] >= 0
    again["tmax"][m] = hits["t"][m] * np.float32(1.0001)  # just beyond the hit: same hit
    h2 = agg.Intersect(again)
    assert (h2["prim"][m] == hits["prim"][m]).all() and (h2["t"][m] == hits["t"][m]).all()
    again["tmax"][m] = hits["t"][m] * np.float32(0.9999)  # just short of the closest hit: miss
    assert (agg.Intersect(again)["prim"][m] == -1).all()
    agg.close()


def test_trace_batches_equals_separate_calls():
    """nnbvh_trace_batches_device (closest + any + any-with-counts traced concurrently on
    internal streams) must give exactly what the separate entry points give."""
    import torch
    from nn_bvh_amd import HIT_DTYPE
    verts, prims = ss.random_soup(5000, 300, 31)
    agg = BVHAggregate(prims, verts)
    ra = scene.random_rays(30000, verts.min(0), verts.max(0), 1)
    rb = scene.random_rays(20000, verts.min(0), verts.max(0), 2, tmax=np.float32(1 - 1e-4))
    exp_a = agg.Intersect(ra)
    exp_b, exp_v, exp_t = agg.IntersectP(rb, counts=True)

    def dev(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()
    da, db = dev(ra), dev(rb)
    oa = torch.zeros(len(ra) * 32, dtype=torch.uint8, device="cuda")
    ob1 = torch.zeros(len(rb), dtype=torch.uint8, device="cuda")
    ob2 = torch.zeros(len(rb), dtype=torch.uint8, device="cuda")
    ov = torch.zeros(len(rb), dtype=torch.int32, device="cuda")
    ot = torch.zeros(len(rb), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):  # repeated use re-records the fork/join events
        agg.trace_batches_device([("closest", da.data_ptr(), len(ra), oa.data_ptr()),
                                  ("any", db.data_ptr(), len(rb), ob1.data_ptr()),
                                  ("any", db.data_ptr(), len(rb), ob2.data_ptr(), ov.data_ptr(),
                                   ot.data_ptr()),
                                  ("closest", da.data_ptr(), 0, oa.data_ptr())], st)
    torch.cuda.synchronize()
    assert oa.cpu().numpy().view(HIT_DTYPE).tobytes() == exp_a.tobytes()
    assert (ob1.cpu().numpy() == exp_b).all() and (ob2.cpu().numpy() == exp_b).all()
    assert (ov.cpu().numpy() == exp_v).all() and (ot.cpu().numpy() == exp_t).all()
    agg.trace_batches_device([], st)
    from nn_bvh_amd import NNBVHError
    with pytest.raises(NNBVHError, match="bad batch"):
        agg.trace_batches_device([("closest", 0, 5, oa.data_ptr())], st)
    agg.close()


def test_fused_batches_one_launch_equals_the_oracle():
    """nnbvh_trace_batches_device with closest / occlusion-only batches runs as ONE launch (mode 3:
    the wavefronts drain the batches one after the other, lanes of one wave may carry rays of
    different batches at the seams): results must be those of the separate entry points = the oracle,
    for ragged batch sizes, host-only primitives, a scene with patches and a two-level scene."""
    import torch
    from nn_bvh_amd import HIT_DTYPE, instancing
    from test_instancing import two_level_scene
    st = torch.cuda.current_stream().cuda_stream

    def dev(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()

    def run(agg, batches):
        outs, args = [], []
        for kind, rays in batches:
            d = dev(rays)
            o = torch.full((max(len(rays), 1) * (32 if kind == "closest" else 1),), 0x5A, dtype=torch.uint8, device="cuda")
            outs.append((kind, d, o))
            args.append((kind, d.data_ptr(), len(rays), o.data_ptr()))
        agg.trace_batches_device(args, st)
        torch.cuda.synchronize()
        return [o.cpu().numpy().view(HIT_DTYPE) if k == "closest" else o.cpu().numpy() for k, _, o in outs]

    # lean scene (no patches): sizes that are no multiple of anything, one tiny, one of a single ray
    verts, prims = ss.random_soup(4000, 0, 41)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    lo, hi = verts.min(0) - 1, verts.max(0) + 1
    batches = [("closest", scene.random_rays(30011, lo, hi, 1)), ("any", scene.random_rays(7, lo, hi, 2, tmax=0.9)),
               ("closest", scene.random_rays(1, lo, hi, 3)), ("any", scene.random_rays(20333, lo, hi, 4, tmax=0.7))]
    got = run(agg, batches)
    for (kind, rays), g in zip(batches, got):
        if kind == "closest":
            assert g.tobytes() == ob.closest(tree.nodes, tree.ordered_prims, verts, rays, 4).tobytes()
        else:
            assert np.array_equal(g[:len(rays)], ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, 4)[0])
    # the multi-stream path gives the same
    agg.set_option("fused_batches", 0)
    again = run(agg, batches)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(got, again))
    agg.close()
    # patches + host-only primitives (general kernels)
    verts, prims = ss.random_soup(3000, 400, 42)
    extra = np.zeros(20, prims.dtype)
    extra["kind"], extra["id"] = 3, len(prims) + np.arange(20)
    allp = np.concatenate([prims, extra])
    rng = np.random.default_rng(5)
    blo = rng.uniform(-8, 8, (len(allp), 3)).astype(np.float32)
    pb = np.concatenate([blo, blo + rng.uniform(0.5, 2, (len(allp), 3)).astype(np.float32)], 1)
    tree = build_tree(allp, verts, prim_bounds=pb)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    batches = [("any", scene.random_rays(9001, verts.min(0) - 2, verts.max(0) + 2, 6)),
               ("closest", scene.random_rays(15013, verts.min(0) - 2, verts.max(0) + 2, 7))]
    got = run(agg, batches)
    assert np.array_equal(got[0], ob.any_hit(tree.nodes, tree.ordered_prims, verts, batches[0][1], 4)[0])
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, batches[1][1], 4)
    assert got[1].tobytes() == exp.tobytes() and (exp["instance"] == -1).any() and (got[0] == 2).any()
    agg.close()
    # two-level scene
    verts, nodes, aprims, instances, n_top, _ = two_level_scene(5, 20)
    agg = BVHAggregate.from_tree(nodes, aprims, verts, instances=instances, n_top_nodes=n_top)
    ra = scene.random_rays(12007, [-30, -30, -30], [30, 30, 30], 8)
    rb = scene.random_rays(8003, [-30, -30, -30], [30, 30, 30], 9, tmax=0.8)
    got = run(agg, [("closest", ra), ("any", rb)])
    assert got[0].tobytes() == ob.closest_inst(nodes, aprims, verts, instances, ra, 4).tobytes()
    assert np.array_equal(got[1], ob.any_hit_inst(nodes, aprims, verts, instances, rb, 4)[0])
    agg.close()
    del instancing


def test_concurrent_host_threads_and_streams():
    """The ABI's thread-safety contract: concurrent calls from several host threads on one
    scene, and concurrent device-pointer launches on several streams, give each call exactly
    its own serial result."""
    import threading
    import torch
    from nn_bvh_amd import HIT_DTYPE
    verts, prims = ss.random_soup(6000, 0, 41)
    agg = BVHAggregate(prims, verts)
    batches = [scene.random_rays(15000 + 1000 * k, verts.min(0), verts.max(0), 100 + k) for k in range(6)]
    expected = [agg.Intersect(b) for b in batches]
    got = [None] * len(batches)

    def work(k):
        for _ in range(3):
            got[k] = agg.Intersect(batches[k])

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(len(batches)):
        assert got[k].tobytes() == expected[k].tobytes(), f"thread {k}"
    # device path: one stream per batch, all in flight together
    streams = [torch.cuda.Stream() for _ in batches]
    d_in = [torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).cuda() for b in batches]
    d_out = [torch.zeros(len(b) * 32, dtype=torch.uint8, device="cuda") for b in batches]
    torch.cuda.synchronize()
    for rep in range(3):
        for k, st in enumerate(streams):
            agg.intersect_device(d_in[k].data_ptr(), d_out[k].data_ptr(), len(batches[k]), st.cuda_stream)
    torch.cuda.synchronize()
    for k in range(len(batches)):
        assert d_out[k].cpu().numpy().view(HIT_DTYPE).tobytes() == expected[k].tobytes(), f"stream {k}"
    agg.close()


def chain_tree(depth, rng):
    """A maximally skewed tree: every interior node has one leaf child (first) and one interior
    child (second), `depth` edges deep — built by hand in the LinearBVHNode layout."""
    from nn_bvh_amd import NODE_DTYPE, make_prims
    n_leaves = depth + 1
    x = np.arange(n_leaves, dtype=np.float32) * 2.0
    c = np.stack([x, np.zeros_like(x), np.zeros_like(x)], 1)[:, None, :]
    verts = (c + rng.uniform(-0.7, 0.7, size=(n_leaves, 3, 3))).reshape(-1, 3).astype(np.float32)
    prims = make_prims(np.arange(3 * n_leaves, dtype=np.int32).reshape(n_leaves, 3))
    nodes = np.zeros(2 * n_leaves - 1, NODE_DTYPE)
    lo = verts.reshape(n_leaves, 3, 3).min(1)
    hi = verts.reshape(n_leaves, 3, 3).max(1)
    # node 2k = interior k (children: leaf 2k+1, interior/leaf 2k+2); last node = last leaf
    for k in range(n_leaves - 1):
        nodes[2 * k]["pmin"], nodes[2 * k]["pmax"] = lo[k:].min(0), hi[k:].max(0)
        nodes[2 * k]["offset"], nodes[2 * k]["nprims"], nodes[2 * k]["axis"] = 2 * k + 2, 0, 0
        nodes[2 * k + 1]["pmin"], nodes[2 * k + 1]["pmax"] = lo[k], hi[k]
        nodes[2 * k + 1]["offset"], nodes[2 * k + 1]["nprims"] = k, 1
    nodes[-1]["pmin"], nodes[-1]["pmax"] = lo[-1], hi[-1]
    nodes[-1]["offset"], nodes[-1]["nprims"] = n_leaves - 1, 1
    return verts, prims, nodes


def test_maximum_depth_and_maximum_leaf_size():
    """The limits of the reference's data structures: a 64-deep tree (its nodesToVisit[64],
    aggregates.cpp:538) is accepted and exact, a deeper one is refused; a leaf with 65 535
    primitives (uint16 nPrimitives, aggregates.cpp:511) is traversed exactly."""
    from nn_bvh_amd import NNBVHError
    rng = np.random.default_rng(3)
    verts, prims, nodes = chain_tree(64, rng)
    agg = BVHAggregate.from_tree(nodes, prims, verts)
    assert agg.info["depth"] == 64
    o = np.stack([np.full(3000, 140.0), rng.uniform(-0.5, 0.5, 3000), rng.uniform(-0.5, 0.5, 3000)], 1)
    d = np.stack([-np.ones(3000), rng.uniform(-2e-3, 2e-3, 3000), rng.uniform(-2e-3, 2e-3, 3000)], 1)
    rays = np.concatenate([make_rays(o, d), scene.random_rays(3000, verts.min(0) - 1, verts.max(0) + 1, 2)])
    for w in (4, 8, 16):
        agg.set_option("stack_window", w)
        assert_hits_equal(agg.Intersect(rays), ob.closest(nodes, prims, verts, rays), f"depth 64, window {w}")
    assert ob.closest(nodes, prims, verts, rays)["nodes_visited"].max() > 100
    agg.close()
    v2, p2, n2 = chain_tree(65, rng)
    with pytest.raises(NNBVHError, match="deeper than the 64-entry"):
        BVHAggregate.from_tree(n2, p2, v2)
    # one leaf of 65 535 primitives
    n = 65535
    c = np.array([0.5, 0.5, 0.5], np.float32)
    h = (rng.integers(1, 64, size=(n, 3)) / 128.0).astype(np.float32)
    w3 = (rng.integers(-63, 64, size=(n, 3)) / 64.0).astype(np.float32) * h
    verts = np.stack([c - h, c + h, c + w3], 1).reshape(-1, 3).astype(np.float32)
    from nn_bvh_amd import make_prims
    prims = make_prims(np.arange(3 * n, dtype=np.int32).reshape(n, 3))
    tree = build_tree(prims, verts)
    assert len(tree.nodes) == 1 and tree.nodes["nprims"][0] == 65535
    rays = scene.random_rays(600, [-0.5] * 3, [1.5] * 3, 4)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, 16)
    assert_hits_equal(agg.Intersect(rays), exp, "65535-prim leaf")
    assert exp["prim_tests"].max() == 65535
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    eocc, evis, etst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, 16)
    assert (occ == eocc).all() and (tst == etst).all()
    agg.close()


def test_device_entry_points_are_hip_graph_capturable():
    """The *_device entry points only enqueue work on the caller's stream (a memset and kernels)
    once that stream's workspace exists, so a wavefront iteration can be captured in a hipGraph and
    replayed; results equal the eager ones."""
    import torch
    verts, prims = ss.random_soup(3000, 300, 41)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    rays = scene.random_rays(50000, verts.min(0) - 2, verts.max(0) + 2, 42)
    shadow = rays.copy()
    shadow["tmax"] = np.float32(7.5)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)  # noqa: E731
    d_rays, d_shadow = up(rays), up(shadow)
    d_hits = torch.zeros(len(rays) * 32, dtype=torch.uint8, device=dev)
    d_occ = torch.zeros(len(rays), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()  # the buffers' zero-fills ran on the default stream

    def step(stream):
        agg.intersect_device(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), stream)
        agg.intersect_p_device(d_shadow.data_ptr(), d_occ.data_ptr(), len(rays), stream=stream)

    with torch.cuda.stream(side):
        step(side.cuda_stream)  # warm-up: creates this stream's workspace (allocation is not capturable)
    torch.cuda.synchronize()
    eager_hits, eager_occ = d_hits.clone(), d_occ.clone()
    d_hits.zero_()
    d_occ.zero_()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step(side.cuda_stream)
    torch.cuda.synchronize()
    assert not d_hits.any(), "capture must not execute the work"
    for _ in range(3):
        d_hits.zero_()
        d_occ.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d_hits, eager_hits) and torch.equal(d_occ, eager_occ)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    assert_hits_equal(d_hits.cpu().numpy().view(exp.dtype), exp, "graph replay")
    # the one-launch form of the same iteration (mode-3 kernel) captures and replays as well
    def fused(stream):
        agg.trace_batches_device([("closest", d_rays.data_ptr(), len(rays), d_hits.data_ptr()),
                                  ("any", d_shadow.data_ptr(), len(rays), d_occ.data_ptr())], stream)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):
        fused(side.cuda_stream)
    d_hits.zero_()
    d_occ.zero_()
    torch.cuda.synchronize()
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(d_hits, eager_hits) and torch.equal(d_occ, eager_occ)
    agg.close()


@pytest.mark.gpu
def test_host_buffer_pipeline_is_independent_of_chunking_and_pinning():
    """nnbvh_intersect_closest / _any cut a host batch into chunks that rotate over three streams; buffers the
    caller pinned are DMA'd directly, pageable ones are staged.  Whatever the chunk size (13 chunks, ragged last
    chunk, one chunk) and wherever the buffers live, the records are the oracle's."""
    # 52 048 rays: host_chunk 4096 -> 6 chunks of 8 675; 20 000 -> 3 chunks, the last ragged; 2^20 -> one chunk
    import ctypes
    from nn_bvh_amd import _lib
    verts, prims = ss.random_soup(4000, 300, 41)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    rays = np.concatenate([scene.random_rays(50_000, verts.min(0) - 2, verts.max(0) + 2, 42), ss.edge_case_rays(verts, prims, 43)])
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    eocc, evis, etst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    L = _lib.lib()
    # buffers to pin own their pages (anonymous mappings): a registration covers whole pages, and heap pages shared
    # with other arrays must not stay mapped into the GPU's address space (include/nnbvh.h, nnbvh_host_register)
    import mmap
    maps = [mmap.mmap(-1, (len(rays) * 32 + 4095) // 4096 * 4096) for _ in range(2)]
    pinned_rays = np.frombuffer(maps[0], rays.dtype, len(rays))
    pinned_hits = np.frombuffer(maps[1], exp.dtype, len(rays))
    pinned_rays[:] = rays
    assert L.nnbvh_host_register(ctypes.c_void_p(rays.ctypes.data + 32), 4096) != 0 and "page-aligned" in _lib.last_error()
    for a, m in zip((pinned_rays, pinned_hits), maps):
        assert a.ctypes.data % 4096 == 0
        assert L.nnbvh_host_register(ctypes.c_void_p(a.ctypes.data), len(m)) == 0, _lib.last_error()
    try:
        for chunk in (4096, 20000, 1 << 20):
            agg.set_option("host_chunk", chunk)
            assert agg.Intersect(rays).tobytes() == exp.tobytes(), f"pageable, chunk {chunk}"
            pinned_hits[:] = 0
            assert agg.Intersect(pinned_rays, pinned_hits).tobytes() == exp.tobytes(), f"pinned, chunk {chunk}"
            occ, vis, tst = agg.IntersectP(rays, counts=True)
            assert np.array_equal(occ, eocc) and np.array_equal(vis, evis) and np.array_equal(tst, etst)
            assert np.array_equal(agg.IntersectP(pinned_rays), eocc)
        # a single ray (the per-ray adapter's shape) and an empty batch
        assert agg.Intersect(rays[:1]).tobytes() == exp[:1].tobytes() and len(agg.Intersect(rays[:0])) == 0
    finally:
        for a in (pinned_rays, pinned_hits):
            assert L.nnbvh_host_unregister(ctypes.c_void_p(a.ctypes.data)) == 0, _lib.last_error()
    agg.close()
    del a, pinned_rays, pinned_hits  # the mappings go away with their last reference
