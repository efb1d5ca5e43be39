"""IntersectShadowTr / IntersectOneRandom (wavefront/aggregate.cpp:70-116) in their media-free form.

The oracle is composed in Python from pieces that are each pinned to the compiled reference:
BVHAggregate::Intersect (oracle traversal), Triangle::InteractionFromIntersection (pi, n),
SpawnRayTo / OffsetRayOrigin, Hash and the WeightedReservoirSampler on PCG32 (tests/
test_aux_oracle.py, tests/test_interaction.py).  The device entry points must reproduce the per-item
verdicts, the selected hits, the reservoir probabilities and the pixel radiance bit for bit."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import HIT_DTYPE, RAY_DTYPE, build_tree, scene


def layered_scene(seed=3):
    """Parallel jittered sheets of triangles (so that segments cross several surfaces) + a soup."""
    rng = np.random.default_rng(seed)
    verts, tris = [], []
    for k, z in enumerate(np.linspace(-3, 3, 7)):
        n = 10
        x, y = np.meshgrid(np.linspace(-4, 4, n + 1), np.linspace(-4, 4, n + 1), indexing="ij")
        v = np.stack([x, y, z + 0.15 * rng.standard_normal(x.shape)], -1).reshape(-1, 3)
        idx = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
        a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
        t = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
        keep = rng.random(len(t)) < 0.8  # holes
        tris.append(t[keep] + sum(len(w) for w in verts))
        verts.append(v)
    verts = np.concatenate(verts).astype(np.float32)
    tris = np.concatenate(tris).astype(np.int32)
    return verts, tris


def oracle_interactions(verts, tris, rays, hits):
    """pi low / high and n of Triangle::InteractionFromIntersection for triangle hits (no uv / n / s arrays)."""
    rec = np.zeros((len(hits), 45), np.float32)
    rec[:, 0:9] = verts[tris[hits["prim"]]].reshape(-1, 9)
    rec[:, 9], rec[:, 10], rec[:, 11] = hits["b0"], hits["b1"], hits["b2"]
    rec[:, 12:15] = -rays["d"]
    rec[:, 18] = rays["time"]
    out = ob.triangle_interaction_batch(rec)
    return out[:, 38:41], out[:, 41:44], out[:, 11:14]


def oracle_shadow_tr(tree, verts, tris, rays, prim_class, Ld, ru, rl, pixel, L):
    n = len(rays)
    state = np.zeros(n, np.uint8)
    p_light = (rays["o"] + rays["d"] * rays["tmax"][:, None]).astype(np.float32)
    cur, idx = rays.copy(), np.arange(n)
    passes = 0
    while len(idx):
        passes += 1
        live = (cur["d"] != 0).any(1)
        cur, idx = cur[live], idx[live]
        h = ob.closest(tree.nodes, tree.ordered_prims, verts, cur)
        hit = h["prim"] >= 0
        iface = hit & ((prim_class[np.maximum(h["prim"], 0)] & 2) != 0)
        state[idx[hit & ~iface]] = 1
        cur, idx, h = cur[iface], idx[iface], h[iface]
        if not len(idx):
            break
        lo, hi, nn = oracle_interactions(verts, tris, cur, h)
        sp = ob.offset_batch(np.concatenate([lo, hi, nn, p_light[idx]], 1))
        cur = cur.copy()
        cur["o"], cur["d"] = sp[:, 3:6], sp[:, 6:9]
    for i in np.nonzero(state == 0)[0]:  # intersect.h:258-273 with T_ray = r_u = r_l = 1
        s = ru[i] * np.float32(1) + rl[i] * np.float32(1)
        acc = s[0]
        for k in (1, 2, 3):
            acc = np.float32(acc + s[k])
        kk = np.float32(1) / np.float32(acc / np.float32(4))
        L[pixel[i]] = L[pixel[i]] + Ld[i] * kk
    return state, passes


def oracle_one_random(tree, verts, tris, p0, p1, material, prim_material):
    n = len(p0)
    sel_hit = np.zeros(n, HIT_DTYPE)
    sel_hit["prim"] = -1
    sel_ray = np.zeros(n, RAY_DTYPE)
    seeds = ob.hash_batch(np.concatenate([p0, p1], 1))
    seed64 = seeds[0].astype(np.uint64) | (seeds[1].astype(np.uint64) << np.uint64(32))
    matches = [[] for _ in range(n)]       # (hit, ray) of every matching surface, in order
    lo, hi, nn = p0.copy(), p0.copy(), np.zeros_like(p0)
    idx = np.arange(n)
    while len(idx):
        sp = ob.offset_batch(np.concatenate([lo, hi, nn, p1[idx]], 1))
        rays = np.zeros(len(idx), RAY_DTYPE)
        rays["o"], rays["d"], rays["tmax"] = sp[:, 3:6], sp[:, 6:9], 1.0
        live = (rays["d"] != 0).any(1)
        rays, idx = rays[live], idx[live]
        h = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
        hit = h["prim"] >= 0
        rays, idx, h = rays[hit], idx[hit], h[hit]
        if not len(idx):
            break
        lo, hi, nn = oracle_interactions(verts, tris, rays, h)
        for j, i in enumerate(idx):
            if prim_material[h["prim"][j]] == material[i]:
                matches[i].append((h[j].copy(), rays[j].copy()))
    pdf, wsum = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for i in range(n):
        k = len(matches[i])
        # the reservoir over k unit-weight candidates, seeded with Hash(p0, p1)
        rec = np.concatenate([p0[i], p1[i], [k]]).astype(np.float32)[None]
        sel, out = ob.wrs_batch(rec)
        pdf[i], wsum[i] = out[0]
        if k:
            sel_hit[i], sel_ray[i] = matches[i][sel[0]]
    return sel_hit, sel_ray, pdf, wsum, seed64


def test_python_oracle_pieces_are_consistent():
    """CPU sanity of the composed oracle: arriving rays on an all-interface scene = all rays."""
    verts, tris = layered_scene()
    prims = ss.make_prims(tris)
    tree = build_tree(prims, verts)
    rays = scene.random_rays(300, [-3, -3, -5], [3, 3, 5], 4, tmax=1 - 1e-4)
    cls = np.full(len(tris), 2, np.uint8)
    L = np.zeros((300, 4), np.float32)
    one = np.ones((300, 4), np.float32)
    state, passes = oracle_shadow_tr(tree, verts, tris, rays, cls, one, one, one, np.arange(300), L)
    assert (state == 0).all() and passes > 3 and np.allclose(L, 0.5)  # Ld * 1 / avg(r_u + r_l) = 1 / 2
    cls[:] = 0
    state, passes = oracle_shadow_tr(tree, verts, tris, rays, cls, one, one, one, np.arange(300), L.copy())
    assert passes == 1 and 0.3 < (state == 1).mean() < 1.0


@pytest.mark.gpu
def test_device_shadow_tr_equals_oracle():
    import torch
    from nn_bvh_amd import BVHAggregate
    from nn_bvh_amd.interaction import ShadingMesh
    from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate
    verts, tris = layered_scene()
    prims = ss.make_prims(tris)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    mesh = ShadingMesh(verts, tris)
    rng = np.random.default_rng(8)
    n = 6000
    rays = scene.random_rays(n, [-3.5, -3.5, -5], [3.5, 3.5, 5], 9, tmax=1 - 1e-4)
    rays["d"][::97] = 0          # zero direction: the walk never starts, the ray arrives
    rays["time"] = rng.random(n).astype(np.float32)
    cls = rng.choice(np.array([0, 1, 2, 2, 2, 6], np.uint8), len(tris))  # mostly interface surfaces
    Ld = (rng.random((n, 4), np.float32) * 2).astype(np.float32)
    ru = (rng.random((n, 4), np.float32) + 0.5).astype(np.float32)
    rl = (rng.random((n, 4), np.float32) + 0.5).astype(np.float32)
    pixel = rng.permutation(n).astype(np.int32)
    L0 = rng.random((n, 4), np.float32).astype(np.float32)
    expL = L0.copy()
    exp_state, passes = oracle_shadow_tr(tree, verts, tris, rays, cls, Ld, ru, rl, pixel, expL)
    assert passes >= 4 and 0.05 < (exp_state == 0).mean() < 0.95
    dev = torch.device("cuda", 0)
    wf = WavefrontAggregate(agg, cls)
    q = RayQueue.from_records(rays, dev, shadow=True)
    q.time = torch.from_numpy(np.ascontiguousarray(rays["time"])).to(dev)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    L = t(L0.copy())
    state = torch.full((n,), 77, dtype=torch.uint8, device=dev)
    wf.IntersectShadowTr(n, q, mesh, t(Ld), t(ru), t(rl), t(pixel), L, state)
    torch.cuda.synchronize()
    assert np.array_equal(state.cpu().numpy(), exp_state)
    assert L.cpu().numpy().tobytes() == expL.tobytes()
    # device-side queue size: items beyond it are untouched
    q.size.fill_(n // 2)
    L2 = t(L0.copy())
    state2 = torch.full((n,), 77, dtype=torch.uint8, device=dev)
    wf.IntersectShadowTr(n, q, mesh, t(Ld), t(ru), t(rl), t(pixel), L2, state2)
    expL2 = L0.copy()
    oracle_shadow_tr(tree, verts, tris, rays[: n // 2], cls, Ld, ru, rl, pixel, expL2)
    assert (state2.cpu().numpy()[n // 2:] == 77).all() and L2.cpu().numpy().tobytes() == expL2.tobytes()
    # without interface surfaces it is IntersectShadow with the other rounding of the weight
    wf0 = WavefrontAggregate(agg, np.zeros(len(tris), np.uint8))
    q.size.fill_(n)
    st0 = torch.zeros(n, dtype=torch.uint8, device=dev)
    wf0.IntersectShadowTr(n, q, mesh, t(Ld), t(ru), t(rl), t(pixel), t(L0.copy()), st0)
    occ = agg.IntersectP(rays)
    live = (rays["d"] != 0).any(1)
    assert np.array_equal(st0.cpu().numpy()[live], occ[live])
    agg.close()
    mesh.close()


@pytest.mark.gpu
def test_device_one_random_equals_oracle():
    import torch
    from nn_bvh_amd import BVHAggregate
    from nn_bvh_amd.interaction import ShadingMesh
    from nn_bvh_amd.wavefront import WavefrontAggregate
    verts, tris = layered_scene(5)
    prims = ss.make_prims(tris)
    tree = build_tree(prims, verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    mesh = ShadingMesh(verts, tris)
    rng = np.random.default_rng(10)
    n = 3000
    p0 = rng.uniform([-3, -3, -4.5], [3, 3, 4.5], (n, 3)).astype(np.float32)
    p1 = rng.uniform([-3, -3, -4.5], [3, 3, 4.5], (n, 3)).astype(np.float32)
    p1[::50] = p0[::50]                      # zero-length segments
    prim_material = rng.integers(0, 3, len(tris)).astype(np.int32)
    material = rng.integers(0, 3, n).astype(np.int32)
    eh, er, epdf, ewsum, _ = oracle_one_random(tree, verts, tris, p0, p1, material, prim_material)
    assert (eh["prim"] >= 0).mean() > 0.3 and ewsum.max() >= 3
    dev = torch.device("cuda", 0)
    wf = WavefrontAggregate(agg)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    sh, sr, pdf, wsum = wf.IntersectOneRandom(n, t(p0), t(p1), t(material), mesh, t(prim_material))
    torch.cuda.synchronize()
    gh = sh.cpu().numpy().view(HIT_DTYPE).reshape(-1)
    gr = sr.cpu().numpy().view(RAY_DTYPE).reshape(-1)
    assert np.array_equal(wsum.cpu().numpy().view(np.uint32), ewsum.view(np.uint32))
    assert np.array_equal(pdf.cpu().numpy().view(np.uint32), epdf.view(np.uint32))
    assert np.array_equal(gh["prim"], eh["prim"])
    sel = eh["prim"] >= 0
    assert gh[sel].tobytes() == eh[sel].tobytes() and gr[sel].tobytes() == er[sel].tobytes()
    agg.close()
    mesh.close()
