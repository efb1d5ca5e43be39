#!/usr/bin/env python3
"""Times the hit -> SurfaceInteraction post-pass (k_triangle_interactions) on the crown bench step's
primary and bounce hit records.  Usage: python tools/interaction_probe.py [--scene crown] [--spp 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--spp", type=int, default=8)
    args = ap.parse_args()
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    from nn_bvh_amd.interaction import ShadingMesh
    verts, tris, source = scene.load_scene(args.scene)
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    cam = args.scene if args.scene in scene.CAMERAS else "crown"
    primary = np.concatenate([scene.camera_rays(cam, seed=1, sample=s) for s in range(args.spp)])
    hits = agg.Intersect(primary)
    mesh = ShadingMesh(verts, tris)
    n = len(primary)
    d_rays = torch.from_numpy(primary.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.from_numpy(hits.view(np.uint8).reshape(-1)).cuda()
    out = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ts = []
    for rep in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        mesh.interactions_device(d_hits.data_ptr(), n, out.data_ptr(), d_rays=d_rays.data_ptr(), stream=stream)
        b.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    hit = float((hits["prim"] >= 0).mean())
    gb = n * (32 + 16 + hit * (12 + 36 + 192) + (1 - hit) * 16) / 1e9
    print(f"{source}: {n} records ({hit:.0%} hits) in {ms:.3f} ms = {n / ms / 1e3:.0f} M records/s, {gb / ms * 1e3:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
