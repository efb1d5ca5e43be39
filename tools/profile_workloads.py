#!/usr/bin/env python3
"""Workloads for the kernel families bench.py does not reach, sized so that one run takes seconds: run under
tools/profile_bench.sh --script tools/profile_workloads.py <which> for rocprofv3 evidence (profiles/r03_*).
    patches   soup of 400 K triangles + 400 K bilinear patches: PATCH = 1 instances, closest / any / one-launch
    alpha     1 M-triangle soup, 70 % alpha-tested (kinds 4 / 5): ALPHA = 1 instances
    inst      400 placements of one 33 K-triangle killeroo (tools/bench_instances.py's scene): INST = 1
    anim      36 placements of a small mesh, 24 of them AnimatedPrimitives (tests/test_animated.py's scene): INST = 2
    tr        IntersectShadowTr and IntersectOneRandom over 2 M items on sheets of interface surfaces: str_* / or_*
    intr      k_triangle_interactions on 11.2 M crown primary hit records
    crown_primary | crown_bounce | crown_bounce2 | crown_shadow   one ray class of the bench step alone
Prints one line per timed call."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(torch, label, n, fn, reps=5):
    ts = []
    for it in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        if it:
            ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print(f"{label}: {ms:.3f} ms, {n / ms / 1e3:.1f} M items/s", flush=True)
    return ms


def trace_three(torch, agg, rays, label):
    """closest, any (occlusion only), any with counts, and the one-launch form over the same batch"""
    n = len(rays)
    d_r = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_h = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    d_o = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_v = torch.empty(n, dtype=torch.int32, device="cuda")
    d_t = torch.empty(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    timed(torch, f"{label} closest", n, lambda: agg.intersect_device(d_r.data_ptr(), d_h.data_ptr(), n, st))
    timed(torch, f"{label} any", n, lambda: agg.intersect_p_device(d_r.data_ptr(), d_o.data_ptr(), n, stream=st))
    timed(torch, f"{label} any+counts", n, lambda: agg.intersect_p_device(d_r.data_ptr(), d_o.data_ptr(), n, d_v.data_ptr(), d_t.data_ptr(), st))
    try:
        timed(torch, f"{label} one launch (closest + any)", 2 * n,
              lambda: agg.trace_batches_device([("closest", d_r.data_ptr(), n, d_h.data_ptr()), ("any", d_r.data_ptr(), n, d_o.data_ptr())], st))
    except Exception as e:  # the alpha instances have no one-launch form
        print(f"{label} one launch: not available ({e})")
    from nn_bvh_amd import HIT_DTYPE
    h = d_h.cpu().numpy().view(HIT_DTYPE)
    print(f"{label}: hit {np.mean(h['prim'] >= 0):.2f}, V {h['nodes_visited'].mean():.1f}, T {h['prim_tests'].mean():.2f}")


def main():
    which = sys.argv[1]
    import torch
    import scenes_small as ss
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    if which in ("patches", "alpha", "alpha_patch"):
        normals = prim_alpha = None
        if which == "patches":
            verts, prims = ss.random_soup(400_000, 400_000, 31, extent=60.0, size=0.9)
        elif which == "alpha_patch":  # the "patches" soup with 70 % of the patches alpha-tested (kinds 8 .. 11)
            verts, prims = ss.random_soup(400_000, 400_000, 31, extent=60.0, size=0.9)
            rng = np.random.default_rng(35)
            prims = prims.copy()
            pk = rng.choice(np.array([1, 8, 9, 10, 11], np.int32), len(prims), p=[0.3, 0.175, 0.175, 0.175, 0.175])
            prims["kind"] = np.where(prims["kind"] == 1, pk, prims["kind"])
            prim_alpha = rng.choice(np.array([0.0, 0.25, 0.5, 0.9, 1.0], np.float32), len(prims))
            normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
        else:
            verts, prims = ss.random_soup(1_000_000, 0, 32, extent=60.0, size=0.9)
            rng = np.random.default_rng(33)
            prims = prims.copy()
            kinds = rng.choice(np.array([0, 4, 5], np.int32), len(prims), p=[0.3, 0.4, 0.3])
            alpha = rng.choice(np.array([0.0, 0.25, 0.5, 0.9, 1.0], np.float32), len(prims))
            prims["kind"] = kinds
            prims["v"][:, 3] = np.where(kinds == 0, 0, alpha.view(np.int32))
        tree = build_tree(prims, verts)
        if prim_alpha is not None:
            prim_alpha = prim_alpha[tree.ordered_prims["id"]]
        agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, normals=normals, prim_alpha=prim_alpha)
        rays = scene.random_rays(4_000_000, verts.min(0) - 2, verts.max(0) + 2, 34)
        trace_three(torch, agg, rays, which)
    elif which == "anim":
        import test_animated as ta
        verts, prims, _, _, _, _, anims, oa, placements = ta.animated_scene(4, 36)
        nodes, aprims, instances, n_top = ta.rebuild_with_motion_bounds(verts, prims, placements, anims, oa)
        agg = BVHAggregate.from_tree(nodes, aprims, verts, instances=instances, n_top_nodes=n_top, animated=anims)
        rays = scene.random_rays(4_000_000, [-25, -25, -25], [25, 25, 25], 5)
        rays["time"] = np.random.default_rng(6).uniform(-0.2, 1.2, len(rays)).astype(np.float32)
        trace_three(torch, agg, rays, which)
    elif which == "inst":
        from nn_bvh_amd import instancing
        side = 20
        verts, tris = scene.load_blob("killeroos")
        k1 = tris[4:4 + 33264]
        used, inv = np.unique(k1, return_inverse=True)
        kv = verts[used]
        ktris = inv.reshape(-1, 3).astype(np.int32)
        ext = kv.max(0) - kv.min(0)
        ground = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32) * side * ext.max() + [0, 0, kv[:, 2].min()]
        allv = np.concatenate([kv, ground]).astype(np.float32)
        obj = make_prims(ktris)
        top = make_prims(np.array([[0, 1, 2], [2, 3, 0]], np.int32) + len(kv))
        top["id"] += 10_000_000
        rng = np.random.default_rng(1)
        placements = []
        for i in range(side):
            for j in range(side):
                a, s = rng.uniform(0, 2 * np.pi), rng.uniform(0.6, 1.2)
                M = np.eye(4)
                M[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * s
                M[:3, 3] = [(i - side / 2) * ext[0] * 1.3, (j - side / 2) * ext[1] * 1.3, 0]
                placements.append((0, M[:3].astype(np.float32).reshape(12), np.linalg.inv(M)[:3].astype(np.float32).reshape(12)))
        nodes, prims, instances, n_top = instancing.assemble_two_level(top, allv, [obj], placements)
        agg = BVHAggregate.from_tree(nodes, prims, allv, instances=instances, n_top_nodes=n_top)
        c = np.array([0, 0, kv[:, 2].mean()])
        eye = c + np.array([1.0, 0.8, 0.5]) * side * ext.max() * 0.9
        cam = (tuple(eye), tuple(c), (0, 0, 1), 45.0, 1400, 1000)
        rays = np.concatenate([scene.camera_rays(cam, seed=1, sample=s) for s in range(3)])
        trace_three(torch, agg, rays, which)
    elif which == "tr":
        import test_wavefront_tr as tw
        from nn_bvh_amd.interaction import ShadingMesh
        from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate
        verts, tris = tw.layered_scene()
        tree = build_tree(make_prims(tris), verts)
        agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
        mesh = ShadingMesh(verts, tris)
        rng = np.random.default_rng(8)
        n = 2_000_000
        rays = scene.random_rays(n, [-3.5, -3.5, -5], [3.5, 3.5, 5], 9, tmax=1 - 1e-4)
        cls = rng.choice(np.array([0, 1, 2, 2, 2, 6], np.uint8), len(tris))
        dev = torch.device("cuda", 0)
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        Ld, ru, rl = (t((rng.random((n, 4), np.float32) + 0.5).astype(np.float32)) for _ in range(3))
        pixel = t(rng.permutation(n).astype(np.int32))
        L = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        state = torch.zeros(n, dtype=torch.uint8, device=dev)
        wf = WavefrontAggregate(agg, cls)
        q = RayQueue.from_records(rays, dev, shadow=True)
        q.time = t(np.zeros(n, np.float32))
        timed(torch, "IntersectShadowTr", n, lambda: wf.IntersectShadowTr(n, q, mesh, Ld, ru, rl, pixel, L, state), reps=3)
        p0 = t(rng.uniform([-3.5, -3.5, -5], [3.5, 3.5, 5], (n, 3)).astype(np.float32))
        p1 = t(rng.uniform([-3.5, -3.5, -5], [3.5, 3.5, 5], (n, 3)).astype(np.float32))
        material = t(rng.integers(0, 3, n).astype(np.int32))
        prim_mat = t(rng.integers(0, 3, len(tris)).astype(np.int32))
        timed(torch, "IntersectOneRandom", n, lambda: wf.IntersectOneRandom(n, p0, p1, material, mesh, prim_mat), reps=3)
    elif which.startswith("crown_"):
        # one ray class of the crown step alone, as ONE-batch launches of the one-launch kernel (trace_kernel<3>): the
        # batches are generated with the separate-launch kernels (<0>), so the <3> rows of a profile are this class only
        cls = which[len("crown_"):]
        verts, tris, source = scene.load_scene("crown")
        tree = build_tree(make_prims(tris), verts)
        agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
        _, px, py = scene.camera_rays("crown", seed=1, sample=0, return_pixels=True)
        tiles = np.lexsort((px, py, px // 4, py // 4))
        primary = np.stack([scene.camera_rays("crown", seed=1, sample=s) for s in range(8)], 1)[tiles].reshape(-1)
        hits = agg.Intersect(primary)
        rays, kind = primary, "closest"
        if cls in ("bounce", "bounce2"):
            rays = scene.bounce_rays(primary, hits, verts, tris, seed=2)
            if cls == "bounce2":
                rays = scene.bounce_rays(rays, agg.Intersect(rays), verts, tris, seed=4)
        elif cls == "shadow":
            rays, kind = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3), "any"
        n = len(rays)
        d_r = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
        d_o = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        timed(torch, f"crown {cls} ({kind}), one-batch launches of trace_kernel<3>", n,
              lambda: agg.trace_batches_device([(kind, d_r.data_ptr(), n, d_o.data_ptr())], st), reps=6)
    elif which == "intr":
        sys.argv = [sys.argv[0]]
        import interaction_probe
        interaction_probe.main()
    else:
        sys.exit(f"unknown workload {which}")


if __name__ == "__main__":
    main()
