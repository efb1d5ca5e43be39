#!/usr/bin/env python3
"""Does a host<->device copy overlap a running trace kernel on this box?  Times an H2D copy of 358 MB from
pinned memory, an 11.2 M-ray trace, and both at once on two streams (and the same with D2H).  The persistent
trace kernel fills every CU: a copy done by a blit KERNEL has to wait for CUs, a copy done by an SDMA engine
does not."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    verts, tris, source = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    primary = np.concatenate([scene.camera_rays("crown", seed=1, sample=s) for s in range(8)])
    n = len(primary)
    host = torch.from_numpy(primary.view(np.uint8).reshape(-1)).pin_memory()
    d_rays = host.cuda()
    d_copy = torch.empty_like(d_rays)
    h_back = torch.empty_like(host).pin_memory()
    d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    s_k, s_c = torch.cuda.Stream(), torch.cuda.Stream()

    def run(kernel, h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kernel:
            agg.intersect_device(d_rays.data_ptr(), d_hits.data_ptr(), n, s_k.cuda_stream)
        with torch.cuda.stream(s_c):
            if h2d:
                d_copy.copy_(host, non_blocking=True)
            if d2h:
                h_back.copy_(d_copy, non_blocking=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for label, args in (("trace alone", (1, 0, 0)), ("H2D alone", (0, 1, 0)), ("D2H alone", (0, 0, 1)),
                        ("trace + H2D", (1, 1, 0)), ("trace + D2H", (1, 0, 1)), ("H2D + D2H (one stream)", (0, 1, 1)),
                        ("trace + H2D + D2H", (1, 1, 1))):
        run(*args)
        ts = [run(*args) for _ in range(5)]
        print(f"{label:26s} {np.median(ts):7.2f} ms", flush=True)
    print(f"({n} rays = {n * 32 / 1e6:.0f} MB each way; env HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA')})")


if __name__ == "__main__":
    main()
