#!/usr/bin/env python3
"""Ceiling of lane packing: the bounce rays of the crown step traced (a) as they are, (b) every ray
repeated 64 times in a row, so that all 64 lanes of a wavefront carry the same ray: every step of every
wavefront then runs with 64 of 64 lanes and one cache line per fetch.  (b) / (a) bounds what ANY
re-packing of rays by state could buy."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    verts, tris, source = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    primary = scene.camera_rays("crown", seed=1, sample=0)
    hits = agg.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    stream = torch.cuda.current_stream().cuda_stream
    n = (len(bounce) // 64) * 64
    for label, rays in (("primary", primary[:n]), ("bounce", bounce[:n])):
        sub = rays[:: 64][: n // 64]
        for name, batch in ((f"{label}: as they are", rays), (f"{label}: each of {len(sub)} rays x 64 lanes", np.repeat(sub, 64))):
            d = torch.from_numpy(np.ascontiguousarray(batch).view(np.uint8).reshape(-1)).cuda()
            out = torch.empty(len(batch) * 32, dtype=torch.uint8, device="cuda")
            ts = []
            for rep in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                agg.intersect_device(d.data_ptr(), out.data_ptr(), len(batch), stream)
                b.record()
                torch.cuda.synchronize()
                if rep:
                    ts.append(a.elapsed_time(b))
            print(f"{name:45s} {len(batch) / np.median(ts) / 1e3:8.1f} Mray/s", flush=True)


if __name__ == "__main__":
    main()
