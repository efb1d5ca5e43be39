#!/usr/bin/env python3
"""Two-level scene timing: G x G placements of the killeroo mesh (one object definition) over a ground
quad, camera-like rays from above; closest-hit and any-hit Mray/s through the INST kernels."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=16)
    ap.add_argument("--rays", type=int, default=4_000_000)
    args = ap.parse_args()
    import torch
    from nn_bvh_amd import BVHAggregate, instancing, make_prims, make_rays, scene
    verts, tris, source = scene.load_scene("killeroos")
    half = len(tris) // 2  # the first killeroo of the pair
    obj = make_prims(tris[:half])
    lo, hi = verts[np.unique(tris[:half])].min(0), verts[np.unique(tris[:half])].max(0)
    ext = (hi - lo).max() * 1.2
    placements = []
    for gx in range(args.grid):
        for gy in range(args.grid):
            m = np.eye(4)[:3].copy()
            m[:, 3] = [gx * ext, gy * ext, 0.0]
            mi = m.copy()
            mi[:, 3] = -m[:, 3]
            placements.append((0, m.reshape(12).astype(np.float32), mi.reshape(12).astype(np.float32)))
    top = make_prims(np.zeros((0, 3), np.int32))
    nodes, prims, instances, n_top = instancing.assemble_two_level(top, verts, [obj], placements)
    agg = BVHAggregate.from_tree(nodes, prims, verts, instances=instances, n_top_nodes=n_top)
    rng = np.random.default_rng(1)
    span = args.grid * ext
    o = np.stack([rng.uniform(lo[0], lo[0] + span, args.rays), rng.uniform(lo[1], lo[1] + span, args.rays),
                  np.full(args.rays, hi[2] + ext)], 1).astype(np.float32)
    d = np.stack([rng.normal(0, 0.15, args.rays), rng.normal(0, 0.15, args.rays), -np.ones(args.rays)], 1).astype(np.float32)
    rays = make_rays(o, d)
    order = np.lexsort((o[:, 0], (o[:, 1] / ext * 8).astype(np.int64)))  # rows of tiles: a coherent order
    rays = rays[order]
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    out = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for name, fn in (("closest", lambda: agg.intersect_device(d_rays.data_ptr(), out.data_ptr(), len(rays), stream)),
                     ("any", lambda: agg.intersect_p_device(d_rays.data_ptr(), out.data_ptr(), len(rays), stream=stream))):
        ts = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(a.elapsed_time(b))
        print(f"{args.grid}x{args.grid} instances of {half} triangles, {len(rays)} rays, {name}: "
              f"{len(rays) / np.median(ts) / 1e3:.1f} Mray/s", flush=True)
    hits = agg.Intersect(rays[:200000])
    print(f"hit rate {(hits['prim'] >= 0).mean():.2f}, mean nodes visited {hits['nodes_visited'].mean():.1f}")


if __name__ == "__main__":
    main()
