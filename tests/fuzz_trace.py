#!/usr/bin/env python3
"""Randomised device-vs-oracle traversal check (not collected by pytest: run it on a GPU box,
`python tests/fuzz_trace.py --iterations 200`).  Every iteration draws a small scene (triangle soup with
or without bilinear patches, connected height-field mesh, coincident-centroid leaves; snapped to a
coarse grid half of the time so that vertices, box planes and ray origins coincide exactly), a split
method, a leaf size and a ray set that mixes random rays, the parity tests' edge-case rays, rays
starting ON vertices with zero / negative-zero / infinite direction components, rays aimed exactly at
vertices with tMax = the exact distance, and NaN / Inf specials — then compares closest hit (record
bits and both counters), any hit with counts, occlusion-only any hit and the one-launch form against
the oracle.  Prints the seed of every mismatch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_binding as ob  # noqa: E402
import scenes_small as ss  # noqa: E402
from nn_bvh_amd import BVHAggregate, build_tree, make_rays, scene  # noqa: E402

ALPHA_SHARE = 0.0  # --alpha: share of scenes with alpha-tested triangles
SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 0.5, np.inf, -np.inf, 1e-30, -1e-30, np.nan], np.float32)


def draw_scene(rng):
    kind = rng.integers(0, 4)
    seed = int(rng.integers(0, 1 << 30))
    if kind == 0:
        verts, prims = ss.random_soup(int(rng.integers(50, 1500)), 0, seed, extent=4.0, size=0.8)
    elif kind == 1:
        verts, prims = ss.random_soup(int(rng.integers(50, 800)), int(rng.integers(1, 200)), seed, extent=4.0, size=0.8)
    elif kind == 2:
        verts, prims = ss.grid_mesh(int(rng.integers(3, 24)), seed, bump=float(rng.choice([0.0, 0.25, 1.0])))
    else:
        verts, prims = ss.coincident_centroids(int(rng.integers(5, 150)), seed)
    if rng.random() < 0.5:  # snap to a coarse grid: coincident planes, flat boxes, degenerate triangles
        verts = (np.round(verts * 4) / 4).astype(np.float32)
    if ALPHA_SHARE and rng.random() < ALPHA_SHARE:
        # constant-alpha GeometricPrimitives (kinds 4 / 5, cpu/primitive.cpp:50-84) among the triangles
        prims = prims.copy()
        tri = prims["kind"] == 0
        kinds = rng.choice(np.array([0, 4, 5], np.int32), len(prims), p=[0.4, 0.3, 0.3])
        alpha = rng.choice(np.array([0.0, 0.25, 0.5, 0.9, 1.0, 1.5, -0.5], np.float32), len(prims))
        # half of the alpha scenes: some of the meshes are smooth-shaded (kinds 6 / 7, vertex normals in the scene)
        normals = None
        if rng.random() < 0.5:
            kinds = np.where((kinds != 0) & (rng.random(len(prims)) < 0.6), kinds + 2, kinds)
            normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
            normals[rng.random(len(verts)) < 0.05] = 0  # ns falls back to the geometric normal
        prims["kind"] = np.where(tri, kinds, prims["kind"])
        prims["v"][:, 3] = np.where(tri & (kinds != 0), alpha.view(np.int32), prims["v"][:, 3])
        # alpha-tested bilinear patches (kinds 8 .. 11; alpha from the per-primitive array), some of them twisted
        # hard enough for a ray to cross them twice (the re-trace after a rejected hit then finds the patch again)
        patch = prims["kind"] == 1
        prim_alpha = None
        if patch.any() and rng.random() < 0.7:
            pk = rng.choice(np.array([1, 8, 9, 10, 11, 12, 13, 14, 15], np.int32), len(prims))
            prims["kind"] = np.where(patch, pk, prims["kind"])
            prim_alpha = alpha
            if normals is None:
                normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
                normals[rng.random(len(verts)) < 0.05] = 0
            if rng.random() < 0.5:
                verts = verts.copy()
                corner = prims["v"][patch, 3]
                verts[corner] += rng.uniform(-1.0, 1.0, size=(len(corner), 3)).astype(np.float32)
        return verts, prims, normals, prim_alpha
    return verts, prims, None, None


def draw_uvs(rng, verts):
    """(u, v) per vertex for the alpha-tested patches of meshes with uv (kinds 12 .. 15): random, some coincident"""
    uv = rng.random((len(verts), 2)).astype(np.float32)
    uv[rng.random(len(verts)) < 0.08] = np.float32(0.5)
    uv[rng.random(len(verts)) < 0.05, 0] = np.float32(0.25)
    return uv


def draw_rays(rng, verts, prims, n):
    lo, hi = verts.min(0), verts.max(0)
    ext = np.maximum(hi - lo, 1e-3)
    parts = [scene.random_rays(n // 4, lo - 1, hi + 1, int(rng.integers(0, 1 << 30))),
             ss.edge_case_rays(verts, prims, int(rng.integers(0, 1 << 30)), n=max(n // 4, 64))]
    k = n // 4
    o = verts[rng.integers(0, len(verts), k)].copy()
    off = rng.random((k, 3)) < 0.3
    o = np.where(off, o + rng.choice(np.array([-1.5, -0.25, 0.25, 2.0], np.float32), (k, 3)), o).astype(np.float32)
    d = rng.choice(SPECIAL[:9], (k, 3)).astype(np.float32)
    r = make_rays(o, d)
    r["tmax"] = rng.choice(np.array([np.inf, 1.0, 3.0, 0.0, 1e-30, 0.25], np.float32), k)
    parts.append(r)
    # aimed exactly at a vertex, tMax = the distance along the (un-normalised) direction: hits at t == tMax
    a = (lo + rng.random((k, 3)) * ext * np.float32(1.5) - ext * np.float32(0.25)).astype(np.float32)
    if rng.random() < 0.5:
        a = (np.round(a * 4) / 4).astype(np.float32)
    tgt = verts[rng.integers(0, len(verts), k)]
    dd = ((tgt - a) * rng.choice(np.array([1.0, 0.5, 0.25, 2.0], np.float32), (k, 1))).astype(np.float32)
    r = make_rays(a, dd)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r["tmax"] = np.where(rng.random(k) < 0.7, np.nanmax(np.abs((tgt - a) / dd), axis=1), np.inf).astype(np.float32)
    parts.append(r)
    rays = np.concatenate(parts)
    bad = rng.random(len(rays)) < 0.01  # a few NaN / Inf specials anywhere
    for f in ("o", "d"):
        v = rays[f].copy()
        m = bad[:, None] & (rng.random(v.shape) < 0.4)
        v[m] = rng.choice(SPECIAL, int(m.sum()))
        rays[f] = v
    return rays


def compare(agg, tree, verts, rays):
    exp = ob.closest(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    got = agg.Intersect(rays)
    bad = set()
    for f in ("prim", "nodes_visited", "prim_tests", "instance"):  # instance = -1: the record is void (host)
        bad |= set(np.nonzero(got[f] != exp[f])[0].tolist())
    for f in ("t", "b0", "b1", "b2"):  # bit patterns; a NaN equals a NaN (x86 and gfx950 differ in the default NaN's sign)
        differ = (got[f].view(np.uint32) != exp[f].view(np.uint32)) & ~(np.isnan(got[f]) & np.isnan(exp[f]))
        bad |= set(np.nonzero(differ)[0].tolist())
    eocc, evis, etst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays, nthreads=8)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    bad |= set(np.nonzero((occ != eocc) | (vis != evis) | (tst != etst))[0].tolist())
    bad |= set(np.nonzero(agg.IntersectP(rays) != eocc)[0].tolist())
    # the one-launch form: the batch split in a closest-hit and an occlusion-only part, traced by ONE kernel
    import torch
    from nn_bvh_amd._lib import HIT_DTYPE
    half = len(rays) // 2
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.zeros(max(half, 1) * 32, dtype=torch.uint8, device="cuda")
    d_occ = torch.zeros(max(len(rays) - half, 1), dtype=torch.uint8, device="cuda")
    agg.trace_batches_device([("closest", d_rays.data_ptr(), half, d_hits.data_ptr()),
                              ("any", d_rays.data_ptr() + 32 * half, len(rays) - half, d_occ.data_ptr())])
    torch.cuda.synchronize()
    fh = d_hits.cpu().numpy().view(HIT_DTYPE)[:half]
    for f in ("prim", "nodes_visited", "prim_tests"):
        bad |= set(np.nonzero(fh[f] != exp[f][:half])[0].tolist())
    differ = (fh["t"].view(np.uint32) != exp["t"][:half].view(np.uint32)) & ~(np.isnan(fh["t"]) & np.isnan(exp["t"][:half]))
    bad |= set(np.nonzero(differ)[0].tolist())
    bad |= set((half + np.nonzero(d_occ.cpu().numpy()[:len(rays) - half] != eocc[half:])[0]).tolist())
    if bad and os.environ.get("FUZZ_VERBOSE"):
        i = sorted(bad)[0]
        print("  ray", rays[i], rays[i]["o"].view(np.uint32), rays[i]["d"].view(np.uint32))
        print("  oracle closest", exp[i], " device", got[i])
        print("  oracle any", eocc[i], evis[i], etst[i], " device", occ[i], vis[i], tst[i])
    return sorted(bad), exp


def compare_kd(verts, prims, rays, max_prims, normals=None, uvs=None, prim_alpha=None):
    """The same rays through KdTreeAggregate (kd_trace.hip) against the oracle's kd traversal (kd primitives are in
    the caller's order: prim_alpha as drawn; the oracle's attribute arrays are set by the caller)."""
    from nn_bvh_amd.kdtree import KdTreeAggregate, build_kd_tree
    tree = build_kd_tree(prims, verts, max_prims=max_prims)
    agg = KdTreeAggregate.from_tree(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, normals=normals, uvs=uvs,
                                    prim_alpha=prim_alpha)
    exp = ob.kd_closest(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, rays, 8)
    got = agg.Intersect(rays)
    bad = set()
    for f in ("prim", "nodes_visited", "prim_tests"):
        bad |= set(np.nonzero(got[f] != exp[f])[0].tolist())
    for f in ("t", "b0", "b1", "b2"):
        differ = (got[f].view(np.uint32) != exp[f].view(np.uint32)) & ~(np.isnan(got[f]) & np.isnan(exp[f]))
        bad |= set(np.nonzero(differ)[0].tolist())
    eocc, evis, etst = ob.kd_any_hit(tree.nodes, tree.prim_indices, prims, verts, tree.bounds, rays, 8)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    bad |= set(np.nonzero((occ != eocc) | (vis != evis) | (tst != etst))[0].tolist())
    bad |= set(np.nonzero(agg.IntersectP(rays) != eocc)[0].tolist())
    if bad and os.environ.get("FUZZ_VERBOSE"):
        i = sorted(bad)[0]
        print("  kd ray", rays[i], " oracle", exp[i], " device", got[i], " any", eocc[i], evis[i], etst[i], occ[i], vis[i], tst[i])
    agg.close()
    return sorted(bad)


def compare_two_level(seed, n_rays):
    """A random two-level scene (test_instancing.two_level_scene: two object definitions placed with
    random affine transforms + top-level triangles) through the INST kernels against the oracle."""
    from test_instancing import two_level_scene
    rng = np.random.default_rng(seed)
    verts, nodes, prims, instances, n_top, _ = two_level_scene(int(rng.integers(0, 1 << 20)), int(rng.integers(2, 40)))
    lo = np.array([-30, -30, -30.0])
    rays = np.concatenate([scene.random_rays(n_rays // 2, lo, -lo, seed), draw_rays(rng, verts, prims[prims["kind"] != 2], n_rays // 2)])
    agg = BVHAggregate.from_tree(nodes, prims, verts, instances=instances, n_top_nodes=n_top)
    exp = ob.closest_inst(nodes, prims, verts, instances, rays, 8)
    got = agg.Intersect(rays)
    bad = set()
    for f in ("prim", "nodes_visited", "prim_tests", "instance"):
        bad |= set(np.nonzero(got[f] != exp[f])[0].tolist())
    for f in ("t", "b0", "b1", "b2"):
        differ = (got[f].view(np.uint32) != exp[f].view(np.uint32)) & ~(np.isnan(got[f]) & np.isnan(exp[f]))
        bad |= set(np.nonzero(differ)[0].tolist())
    eocc, evis, etst = ob.any_hit_inst(nodes, prims, verts, instances, rays, 8)
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    bad |= set(np.nonzero((occ != eocc) | (vis != evis) | (tst != etst))[0].tolist())
    bad |= set(np.nonzero(agg.IntersectP(rays) != eocc)[0].tolist())
    if bad and os.environ.get("FUZZ_VERBOSE"):
        i = sorted(bad)[0]
        print("  two-level ray", rays[i], " oracle", exp[i], " device", got[i], " any", eocc[i], evis[i], etst[i], occ[i], vis[i], tst[i])
    agg.close()
    return sorted(bad)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kd", action="store_true", help="also trace every scene's kd-tree")
    ap.add_argument("--alpha", type=float, default=0.0, help="share of scenes with alpha-tested triangles")
    ap.add_argument("--two-level", type=int, default=0, metavar="K", help="every K-th iteration also checks a random instanced scene")
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--rays", type=int, default=8000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    global ALPHA_SHARE
    ALPHA_SHARE = args.alpha
    failures = 0
    total = 0
    for it in range(args.iterations):
        seed = args.seed * 100003 + it
        rng = np.random.default_rng(seed)
        verts, prims, normals, prim_alpha = draw_scene(rng)
        split = str(rng.choice(["sah", "hlbvh", "middle", "equal"]))
        max_prims = int(rng.choice([1, 2, 4, 8]))
        tree = build_tree(prims, verts, max_prims, split)
        a_ord = None if prim_alpha is None else prim_alpha[tree.ordered_prims["id"]]
        uvs = None if prim_alpha is None else draw_uvs(rng, verts)
        agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, normals=normals, prim_alpha=a_ord, uvs=uvs)
        ob.set_vertex_normals(normals)
        ob.set_prim_alpha(a_ord)
        ob.set_vertex_uvs(uvs)
        if rng.random() < 0.3:
            agg.set_option("stack_window", int(rng.choice([4, 16])))
        rays = draw_rays(rng, verts, prims, args.rays)
        bad, exp = compare(agg, tree, verts, rays)
        if args.kd:
            ob.set_prim_alpha(prim_alpha)  # kd primitives are in the caller's order
            kbad = compare_kd(verts, prims, rays, max_prims=int(rng.choice([1, 4])), normals=normals, uvs=uvs,
                              prim_alpha=prim_alpha)
            ob.set_prim_alpha(a_ord)
            if kbad:
                print(f"KD MISMATCH seed {seed}: {len(kbad)} rays, first {kbad[:5]}", flush=True)
                bad = bad + kbad
        if args.two_level and it % args.two_level == 0:
            tbad = compare_two_level(seed, args.rays)
            if tbad:
                print(f"TWO-LEVEL MISMATCH seed {seed}: {len(tbad)} rays, first {tbad[:5]}", flush=True)
                bad = bad + tbad
        total += len(rays)
        if bad:
            failures += 1
            print(f"MISMATCH seed {seed}: {len(bad)} of {len(rays)} rays, first {bad[:5]}, scene {len(prims)} prims, "
                  f"{split}/{max_prims}", flush=True)
        elif it % 20 == 0:
            print(f"iteration {it}: {len(prims)} primitives ({split}, leaves of {max_prims}), {len(rays)} rays, "
                  f"{(exp['prim'] >= 0).mean():.0%} hits: equal", flush=True)
        agg.close()
    print(f"{args.iterations} scenes, {total} rays: {failures} scenes with mismatches")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
