#!/usr/bin/env python3
"""Interleaved timing of the one-launch step (nnbvh_trace_batches_device, mode-3 kernel) against the
three separate launches, for combinations of the scheduling knobs; all variants round-robin in ONE
process.  Usage: python tools/fused_probe.py [--primrepeat 1,2] [--intrepeat 3] [--rounds 5]"""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--primrepeat", default="1,2")
    ap.add_argument("--intrepeat", default="3")
    ap.add_argument("--primweight", default="32")
    ap.add_argument("--refill", default="8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--spp", type=int, default=8)
    args = ap.parse_args()
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    verts, tris, source = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    # the bench's ray order: pixel-major, the pixels in 4x4 tiles
    _, px, py = scene.camera_rays("crown", seed=1, sample=0, return_pixels=True)
    tiles = np.lexsort((px, py, px // 4, py // 4))
    primary = np.stack([scene.camera_rays("crown", seed=1, sample=s) for s in range(args.spp)], 1)[tiles].reshape(-1)
    hits = agg.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    dp, db, ds = dev(primary), dev(bounce), dev(shadow)
    o1 = torch.empty(len(primary) * 32, dtype=torch.uint8, device="cuda")
    o2 = torch.empty(len(bounce) * 32, dtype=torch.uint8, device="cuda")
    o3 = torch.empty(len(shadow), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    n = len(primary) + len(bounce) + len(shadow)

    def fused():
        agg.trace_batches_device([("closest", dp.data_ptr(), len(primary), o1.data_ptr()),
                                  ("closest", db.data_ptr(), len(bounce), o2.data_ptr()),
                                  ("any", ds.data_ptr(), len(shadow), o3.data_ptr())], stream)

    def serial():
        agg.intersect_device(dp.data_ptr(), o1.data_ptr(), len(primary), stream)
        agg.intersect_device(db.data_ptr(), o2.data_ptr(), len(bounce), stream)
        agg.intersect_p_device(ds.data_ptr(), o3.data_ptr(), len(shadow), stream=stream)

    combos = list(itertools.product([int(x) for x in args.primrepeat.split(",")],
                                    [int(x) for x in args.intrepeat.split(",")],
                                    [int(x) for x in args.primweight.split(",")],
                                    [int(x) for x in args.refill.split(",")]))
    times = {(c, k): [] for c in combos for k in ("fused", "serial")}
    for rnd in range(args.rounds + 1):
        for c in combos:
            for key, v in (("prim_repeat", c[0]), ("int_repeat", c[1]), ("prim_weight", c[2]), ("refill_weight", c[3])):
                try:
                    agg.set_option(key, v)
                except Exception:  # an older experimental build without the knob
                    pass
            for k, fn in (("fused", fused), ("serial", serial)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                if rnd:
                    times[(c, k)].append(a.elapsed_time(b))
    print(f"# {source}; {n} rays per step")
    print("prim_repeat int_repeat prim_weight refill_weight | one launch ms (Mray/s) | three launches ms (Mray/s)")
    for c in combos:
        f, s = np.median(times[(c, "fused")]), np.median(times[(c, "serial")])
        print(f"{c[0]:11d} {c[1]:10d} {c[2]:11d} {c[3]:13d} | {f:8.3f} ({n / f / 1e3:7.1f}) | {s:8.3f} ({n / s / 1e3:7.1f})", flush=True)


if __name__ == "__main__":
    main()
