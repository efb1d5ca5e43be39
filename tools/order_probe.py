#!/usr/bin/env python3
"""The crown step's 8 spp as 8 concatenated passes (sample-major: each XCD's eighth of the batch is a whole
image) against the same rays pixel-major (the 8 samples of a pixel adjacent: each XCD's eighth is an
eighth of the image).  One-launch step and the three separate launches."""
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
    spp = 8
    passes = [scene.camera_rays("crown", seed=1, sample=s) for s in range(spp)]
    npix = len(passes[0])
    _, px, py = scene.camera_rays("crown", seed=1, sample=0, return_pixels=True)
    pm = np.stack(passes, 1)  # [pixel][sample]

    def tiled(t):
        order = np.lexsort((px, py, px // t, py // t))  # tiles of t x t pixels in row order, scanlines inside
        return pm[order].reshape(-1)

    def morton():
        def spread(v):
            v = v.astype(np.uint64)
            v = (v | (v << 8)) & 0x00FF00FF
            v = (v | (v << 4)) & 0x0F0F0F0F
            v = (v | (v << 2)) & 0x33333333
            v = (v | (v << 1)) & 0x55555555
            return v
        return pm[np.argsort(spread(px) | (spread(py) << 1), kind="stable")].reshape(-1)

    orders = {"sample-major (8 passes concatenated)": np.concatenate(passes),
              "pixel-major (8 samples of a pixel adjacent)": pm.reshape(-1),
              "pixel-major, 16x16 tiles": tiled(16), "pixel-major, 4x4 tiles": tiled(4),
              "pixel-major, Morton order of the pixels": morton()}
    stream = torch.cuda.current_stream().cuda_stream
    for label, primary in orders.items():
        hits = agg.Intersect(primary)
        bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
        shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
        dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()  # noqa: E731
        dp, db, ds = dev(primary), dev(bounce), dev(shadow)
        o1 = torch.empty(len(primary) * 32, dtype=torch.uint8, device="cuda")
        o2 = torch.empty(len(bounce) * 32, dtype=torch.uint8, device="cuda")
        o3 = torch.empty(len(shadow), dtype=torch.uint8, device="cuda")
        n = len(primary) + len(bounce) + len(shadow)

        def fused():
            agg.trace_batches_device([("closest", dp.data_ptr(), len(primary), o1.data_ptr()),
                                      ("closest", db.data_ptr(), len(bounce), o2.data_ptr()),
                                      ("any", ds.data_ptr(), len(shadow), o3.data_ptr())], stream)

        ts = []
        for rep in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fused()
            b.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(a.elapsed_time(b))
        print(f"{label:48s} {n} rays, one launch {np.median(ts):.3f} ms = {n / np.median(ts) / 1e3:.0f} Mray/s", flush=True)


if __name__ == "__main__":
    main()
