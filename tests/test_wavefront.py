"""Wavefront queue interface (reference: wavefront/aggregate.cpp:34-68, wavefront/intersect.h:16-156).

CPU part: the oracle's restatement of the enqueue rules and of RecordShadowRayResult against an
independent numpy formulation.  GPU part: nnbvh_wavefront_* through the Python mirror against the
oracle — hit records bit-equal, queue CONTENTS equal (push order is unspecified in the reference
too: its threads race on WorkQueue::size), pixel radiance bit-equal."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import HIT_DTYPE, _lib, build_tree, scene

QUEUES = _lib.CLOSEST_QUEUES


def rules_numpy(prim, has_medium, prim_class):
    """The enqueue rules written as set algebra (independent of the oracle's loop)."""
    idx = np.arange(len(prim))
    miss, med = prim < 0, has_medium.astype(bool)
    cls = np.where(miss, 0, prim_class[np.maximum(prim, 0)])
    surf = ~miss & ~med
    iface = surf & ((cls & 2) != 0)
    mat = surf & ~iface
    return {"escaped": idx[miss & ~med], "medium_sample": idx[med], "next_ray": idx[iface],
            "hit_area_light": idx[mat & ((cls & 4) != 0)],
            "basic_eval_material": idx[mat & ((cls & 1) == 0)],
            "universal_eval_material": idx[mat & ((cls & 1) != 0)]}


def test_oracle_enqueue_rules():
    rng = np.random.default_rng(1)
    n, n_prims = 5000, 300
    hits = np.zeros(n, HIT_DTYPE)
    hits["prim"] = rng.integers(-1, n_prims, n)
    hits["prim"][rng.random(n) < 0.3] = -1
    has_medium = (rng.random(n) < 0.15).astype(np.uint8)
    prim_class = rng.choice(np.array([0, 1, 2, 4, 5, 6], np.uint8), n_prims)
    got = dict(zip(QUEUES, ob.wavefront_enqueue_closest(hits, has_medium, prim_class)))
    exp = rules_numpy(hits["prim"], has_medium, prim_class)
    for k in QUEUES:
        assert np.array_equal(got[k], exp[k]), k
    # every item lands in exactly one of escaped / medium / next / basic / universal
    one = np.concatenate([got[k] for k in QUEUES if k != "hit_area_light"])
    assert np.array_equal(np.sort(one), np.arange(n))
    # no class table: everything that hits is "basic"
    got = dict(zip(QUEUES, ob.wavefront_enqueue_closest(hits, None, None)))
    assert np.array_equal(got["basic_eval_material"], np.nonzero(hits["prim"] >= 0)[0])
    assert np.array_equal(got["escaped"], np.nonzero(hits["prim"] < 0)[0])


def shadow_inputs(n, n_pixels, seed):
    rng = np.random.default_rng(seed)
    Ld = rng.random((n, 4), np.float32) * 3
    r_u = rng.random((n, 4), np.float32) + np.float32(0.1)
    r_l = rng.random((n, 4), np.float32) + np.float32(0.1)
    px = rng.permutation(n_pixels)[:n].astype(np.int32)  # unique, as in the reference
    L = rng.random((n_pixels, 4), np.float32)
    return Ld, r_u, r_l, px, L


def test_oracle_record_shadow():
    n, n_pixels = 4000, 6000
    Ld, r_u, r_l, px, L = shadow_inputs(n, n_pixels, 2)
    occ = (np.random.default_rng(3).random(n) < 0.4).astype(np.uint8)
    got = ob.record_shadow(occ, Ld, r_u, r_l, px, L)
    s = r_u + r_l
    avg = (((s[:, 0] + s[:, 1]) + s[:, 2]) + s[:, 3]) / np.float32(4)
    exp = L.copy()
    vis = occ == 0
    exp[px[vis]] = L[px[vis]] + Ld[vis] / avg[vis, None]
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    untouched = np.setdiff1d(np.arange(n_pixels), px[vis])
    assert np.array_equal(got[untouched], L[untouched])


# ----------------------------------------------------------------------------------------------
def _setup(seed, n_rays):
    from nn_bvh_amd import BVHAggregate
    from nn_bvh_amd.wavefront import WavefrontAggregate
    verts, prims = ss.random_soup(2500, 400, seed)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    rays = scene.random_rays(n_rays, verts.min(0) - 3, verts.max(0) + 3, seed + 1)
    return verts, prims, tree, agg, rays, WavefrontAggregate


@pytest.mark.gpu
@pytest.mark.parametrize("device_size", [None, 5000, 0])
def test_gpu_intersect_closest_queues(device_size):
    import torch
    from nn_bvh_amd.wavefront import RayQueue, WorkQueue
    max_rays = 7001
    verts, prims, tree, agg, rays, WavefrontAggregate = _setup(21, max_rays)
    n = max_rays if device_size is None else device_size
    rng = np.random.default_rng(5)
    prim_class = rng.choice(np.array([0, 0, 0, 1, 2, 4, 5], np.uint8), len(prims))
    has_medium = (rng.random(max_rays) < 0.1).astype(np.uint8)
    dev = torch.device("cuda", 0)
    rq = RayQueue.from_records(rays, dev)
    rq.has_medium = torch.from_numpy(has_medium).to(dev)
    if device_size is not None:
        rq.size.fill_(device_size)
    wf = WavefrontAggregate(agg, prim_class)
    queues = {k: WorkQueue(max_rays, dev) for k in QUEUES}
    hits_t = torch.full((max_rays, 32), 0xAB, dtype=torch.uint8, device=dev)
    wf.IntersectClosest(max_rays, rq, hits=hits_t, **queues)
    torch.cuda.synchronize()
    hits = hits_t.cpu().numpy().view(HIT_DTYPE).reshape(-1)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays[:n], nthreads=8)
    from test_gpu_parity import assert_hits_equal
    assert_hits_equal(hits[:n], exp, "wavefront closest")
    assert (hits_t[n:].cpu().numpy() == 0xAB).all(), "records beyond the queue size were written"
    expq = dict(zip(QUEUES, ob.wavefront_enqueue_closest(exp, has_medium[:n], prim_class)))
    for k in QUEUES:
        got = np.sort(queues[k].indices().cpu().numpy())
        assert queues[k].Size() == len(expq[k]), k
        assert np.array_equal(got, expq[k]), k
    if n:
        assert len(expq["escaped"]) and len(expq["next_ray"]) and len(expq["hit_area_light"])
    agg.close()


@pytest.mark.gpu
def test_gpu_queue_overflow_and_unwanted_queues():
    import torch
    from nn_bvh_amd.wavefront import RayQueue, WorkQueue
    n = 4096
    verts, prims, tree, agg, rays, WavefrontAggregate = _setup(31, n)
    dev = torch.device("cuda", 0)
    wf = WavefrontAggregate(agg)
    small = WorkQueue(100, dev)
    small.items.fill_(-7)
    esc = WorkQueue(n, dev)
    wf.IntersectClosest(n, RayQueue.from_records(rays, dev), escaped=esc, basic_eval_material=small)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    hit_idx = np.nonzero(exp["prim"] >= 0)[0]
    assert len(hit_idx) > 100
    assert small.Size() == len(hit_idx)          # counted ...
    stored = small.items.cpu().numpy()
    assert len(stored) == 100 and np.isin(stored, hit_idx).all()  # ... but only `capacity` stored
    assert len(np.unique(stored)) == 100
    assert np.array_equal(np.sort(esc.indices().cpu().numpy()), np.nonzero(exp["prim"] < 0)[0])
    agg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device_size", [None, 3000])
def test_gpu_intersect_shadow_records_radiance(device_size):
    import torch
    from nn_bvh_amd.wavefront import RayQueue
    max_rays, n_pixels = 6000, 9000
    verts, prims, tree, agg, rays, WavefrontAggregate = _setup(41, max_rays)
    # finite shadow segments: un-normalised d, tMax = 1 - eps (integrators.h:52-54)
    rays["tmax"] = np.float32(1 - 1e-4)
    rays["d"] *= np.float32(12.0)
    n = max_rays if device_size is None else device_size
    Ld, r_u, r_l, px, L = shadow_inputs(max_rays, n_pixels, 7)
    dev = torch.device("cuda", 0)
    sq = RayQueue.from_records(rays, dev, shadow=True)
    if device_size is not None:
        sq.size.fill_(device_size)
    t = lambda a: torch.from_numpy(a).to(dev)
    L_t, occ_t = t(L), torch.full((max_rays,), 9, dtype=torch.uint8, device=dev)
    wf = WavefrontAggregate(agg)
    wf.IntersectShadow(max_rays, sq, t(Ld), t(r_u), t(r_l), t(px), L_t, occluded=occ_t)
    torch.cuda.synchronize()
    eocc, _, _ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays[:n], nthreads=8)
    assert 0.1 < eocc.mean() < 0.9
    assert np.array_equal(occ_t.cpu().numpy()[:n], eocc)
    assert (occ_t.cpu().numpy()[n:] == 9).all()
    exp = ob.record_shadow(eocc, Ld[:n], r_u[:n], r_l[:n], px[:n], L)
    assert np.array_equal(L_t.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    # without the optional per-ray output
    L2 = t(L)
    wf.IntersectShadow(max_rays, sq, t(Ld), t(r_u), t(r_l), t(px), L2)
    torch.cuda.synchronize()
    assert np.array_equal(L2.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    agg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [(None, None), (5000, 3000), (0, 4000), (7001, 0)])
def test_gpu_closest_and_shadow_in_one_launch(sizes):
    """nnbvh_wavefront_intersect_closest_and_shadow = IntersectShadow of one depth + IntersectClosest of the next in
    one launch (device-side queue sizes per batch): hit records, destination queues, occlusion flags and pixel
    radiance equal to the oracle's — i.e. to the two separate calls — and nothing is written beyond the queue sizes."""
    import torch
    from nn_bvh_amd.wavefront import RayQueue, WorkQueue
    max_rays, max_shadow, n_pixels = 7001, 6000, 9000
    verts, prims, tree, agg, rays, WavefrontAggregate = _setup(51, max_rays)
    srays = scene.random_rays(max_shadow, verts.min(0) - 3, verts.max(0) + 3, 77)
    srays["tmax"] = np.float32(1 - 1e-4)
    srays["d"] *= np.float32(12.0)
    nc = max_rays if sizes[0] is None else sizes[0]
    ns = max_shadow if sizes[1] is None else sizes[1]
    rng = np.random.default_rng(5)
    prim_class = rng.choice(np.array([0, 0, 0, 1, 2, 4, 5], np.uint8), len(prims))
    has_medium = (rng.random(max_rays) < 0.1).astype(np.uint8)
    Ld, r_u, r_l, px, L = shadow_inputs(max_shadow, n_pixels, 7)
    dev = torch.device("cuda", 0)
    rq, sq = RayQueue.from_records(rays, dev), RayQueue.from_records(srays, dev, shadow=True)
    rq.has_medium = torch.from_numpy(has_medium).to(dev)
    if sizes[0] is not None:
        rq.size.fill_(sizes[0])
    if sizes[1] is not None:
        sq.size.fill_(sizes[1])
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    wf = WavefrontAggregate(agg, prim_class)
    queues = {k: WorkQueue(max_rays, dev) for k in QUEUES}
    hits_t = torch.full((max_rays, 32), 0xAB, dtype=torch.uint8, device=dev)
    L_t, occ_t = t(L), torch.full((max_shadow,), 9, dtype=torch.uint8, device=dev)
    wf.IntersectClosestAndShadow(max_rays, rq, max_shadow, sq, t(Ld), t(r_u), t(r_l), t(px), L_t, hits=hits_t,
                                 occluded=occ_t, **queues)
    torch.cuda.synchronize()
    hits = hits_t.cpu().numpy().view(HIT_DTYPE).reshape(-1)
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays[:nc], nthreads=8)
    assert hits[:nc].tobytes() == exp.tobytes()
    assert (hits_t[nc:].cpu().numpy() == 0xAB).all(), "hit records beyond the queue size were written"
    expq = dict(zip(QUEUES, ob.wavefront_enqueue_closest(exp, has_medium[:nc], prim_class)))
    for k in QUEUES:
        assert queues[k].Size() == len(expq[k]), k
        assert np.array_equal(np.sort(queues[k].indices().cpu().numpy()), expq[k]), k
    eocc, _, _ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, srays[:ns], nthreads=8)
    assert np.array_equal(occ_t.cpu().numpy()[:ns], eocc)
    assert (occ_t.cpu().numpy()[ns:] == 9).all(), "occlusion flags beyond the queue size were written"
    expL = ob.record_shadow(eocc, Ld[:ns], r_u[:ns], r_l[:ns], px[:ns], L)
    assert np.array_equal(L_t.cpu().numpy().view(np.uint32), expL.view(np.uint32))
    agg.close()


@pytest.mark.gpu
def test_gpu_wavefront_iteration_is_hip_graph_capturable():
    """One depth of the wavefront loop — queue resets, the shadow queue + the next depth's ray queue in one launch, the
    enqueue and radiance kernels — captured in a hipGraph and replayed: the calls only enqueue kernels on the caller's
    stream once its workspace exists (no gather pass, no allocation)."""
    import torch
    from nn_bvh_amd.wavefront import RayQueue, WorkQueue
    max_rays, max_shadow, n_pixels = 6000, 5000, 8000
    verts, prims, tree, agg, rays, WavefrontAggregate = _setup(61, max_rays)
    srays = scene.random_rays(max_shadow, verts.min(0) - 3, verts.max(0) + 3, 78)
    srays["tmax"] = np.float32(1 - 1e-4)
    srays["d"] *= np.float32(12.0)
    Ld, r_u, r_l, px, L = shadow_inputs(max_shadow, n_pixels, 9)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    rq, sq = RayQueue.from_records(rays, dev), RayQueue.from_records(srays, dev, shadow=True)
    wf = WavefrontAggregate(agg)
    queues = {k: WorkQueue(max_rays, dev) for k in QUEUES}
    hits_t = torch.zeros((max_rays, 32), dtype=torch.uint8, device=dev)
    Ld_t, ru_t, rl_t, px_t, L_t = t(Ld), t(r_u), t(r_l), t(px), t(L)
    L0 = L_t.clone()
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()

    def iteration():
        for q in queues.values():
            q.Reset()
        wf.IntersectClosestAndShadow(max_rays, rq, max_shadow, sq, Ld_t, ru_t, rl_t, px_t, L_t, hits=hits_t, **queues)

    with torch.cuda.stream(side):
        iteration()  # warm-up: creates this stream's workspace (allocation is not capturable)
    torch.cuda.synchronize()
    eager_hits, eager_L = hits_t.clone(), L_t.clone()
    eager_sizes = {k: q.Size() for k, q in queues.items()}
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        iteration()
    for _ in range(2):
        hits_t.zero_()
        L_t.copy_(L0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(hits_t, eager_hits) and torch.equal(L_t, eager_L)
        assert {k: q.Size() for k, q in queues.items()} == eager_sizes
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    assert hits_t.cpu().numpy().view(HIT_DTYPE).reshape(-1).tobytes() == exp.tobytes()
    agg.close()


@pytest.mark.gpu
def test_gpu_wavefront_bad_arguments():
    import ctypes
    verts, prims, tree, agg, rays, _ = _setup(51, 16)
    L = _lib.lib()
    soa = np.zeros(1, _lib.RAY_SOA_DTYPE)  # null coordinate arrays
    q = np.zeros(1, _lib.CLOSEST_QUEUES_DTYPE)
    rc = L.nnbvh_wavefront_intersect_closest(agg._h, 16, _lib.ptr(soa), None, None, 0,
                                             ctypes.c_void_p(16), _lib.ptr(q), None)
    assert rc == 1 and "bad argument" in _lib.last_error()
    rc = L.nnbvh_wavefront_intersect_shadow(agg._h, 16, _lib.ptr(soa), None, None, None, None, None,
                                            None, 0, None, None)
    assert rc == 1
    assert L.nnbvh_wavefront_intersect_closest(agg._h, 0, None, None, None, 0, None, _lib.ptr(q), None) == 0
    agg.close()
