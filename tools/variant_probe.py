#!/usr/bin/env python3
"""A/B of memory layouts (NNBVH_LAYOUT, bvh_layout.cpp) and scheduling knobs on the crown step, all variants
interleaved round-robin in ONE process on one box: per ray class launched alone and the one-launch step.
Every variant's results are compared byte for byte with the first one's.
Usage: python tools/variant_probe.py [--layouts 0,1,2,9,16,17] [--knobs prim_min=0:int_repeat=3,prim_min=8:int_repeat=1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layouts", default="0,1,2,9,16,17")
    ap.add_argument("--knobs", default="prim_min=0:int_repeat=3,prim_min=8:int_repeat=1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--order", default="tile", choices=["tile", "sample"])
    args = ap.parse_args()
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    verts, tris, source = scene.load_scene(args.scene)
    tree = build_tree(make_prims(tris), verts)
    cam = args.scene if args.scene in scene.CAMERAS else "crown"
    _, px, py = scene.camera_rays(cam, seed=1, sample=0, return_pixels=True)
    per = [scene.camera_rays(cam, seed=1, sample=s) for s in range(args.spp)]
    if args.order == "tile":
        tiles = np.lexsort((px, py, px // 4, py // 4))
        primary = np.stack(per, 1)[tiles].reshape(-1)
    else:
        primary = np.concatenate(per)
    aggs = {}
    for lay in [int(x) for x in args.layouts.split(",")]:
        os.environ["NNBVH_LAYOUT"] = str(lay)
        aggs[lay] = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    os.environ.pop("NNBVH_LAYOUT")
    first = next(iter(aggs.values()))
    hits = first.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    if args.scene == "crown":
        shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
    else:
        lo, hi = verts.min(0), verts.max(0)
        shadow = scene.shadow_rays(primary, hits, verts, tris, lo + (hi - lo) * [0.3, 0.9, 0.3],
                                   lo + (hi - lo) * [0.7, 1.0, 0.7], seed=3)
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    dp, db, ds = dev(primary), dev(bounce), dev(shadow)
    mk = lambda n: torch.empty(n, dtype=torch.uint8, device="cuda")  # noqa: E731
    o1, o2, o3 = mk(len(primary) * 32), mk(len(bounce) * 32), mk(len(shadow))
    stream = torch.cuda.current_stream().cuda_stream
    nP, nB, nS = len(primary), len(bounce), len(shadow)
    n = nP + nB + nS
    knobs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in k.split(":")) for k in args.knobs.split(",")]

    def run(agg, kind):
        if kind == "fused":
            agg.trace_batches_device([("closest", dp.data_ptr(), nP, o1.data_ptr()), ("closest", db.data_ptr(), nB, o2.data_ptr()),
                                      ("any", ds.data_ptr(), nS, o3.data_ptr())], stream)
        elif kind == "primary":
            agg.intersect_device(dp.data_ptr(), o1.data_ptr(), nP, stream)
        elif kind == "bounce":
            agg.intersect_device(db.data_ptr(), o2.data_ptr(), nB, stream)
        else:
            agg.intersect_p_device(ds.data_ptr(), o3.data_ptr(), nS, stream=stream)

    kinds = ("fused", "primary", "bounce", "shadow")
    times = {}
    ref = None
    for rnd in range(args.rounds + 1):
        for lay, agg in aggs.items():
            for ki, kn in enumerate(knobs):
                for key, v in kn.items():
                    agg.set_option(key, v)
                for kind in kinds:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    run(agg, kind)
                    b.record()
                    torch.cuda.synchronize()
                    if rnd:
                        times.setdefault((lay, ki, kind), []).append(a.elapsed_time(b))
                if rnd == 0:  # results of the three separate launches (they ran last) against the first variant's
                    cur = (o1.clone(), o2.clone(), o3.clone())
                    if ref is None:
                        ref = cur
                    elif not all(torch.equal(x, y) for x, y in zip(ref, cur)):
                        print(f"!! layout {lay} knobs {kn}: results DIFFER from the first variant", flush=True)
                    run(agg, "fused")
                    torch.cuda.synchronize()
                    if not all(torch.equal(x, y) for x, y in zip(ref, (o1, o2, o3))):
                        print(f"!! layout {lay} knobs {kn}: one-launch results DIFFER", flush=True)
    print(f"# {source}; {n} rays per step ({nP} primary, {nB} bounce, {nS} shadow); order {args.order}; median of {args.rounds}")
    print("layout knobs | step ms (Mray/s) | primary | bounce | shadow  Mray/s")
    for lay in aggs:
        for ki, kn in enumerate(knobs):
            t = {k: float(np.median(times[(lay, ki, k)])) for k in kinds}
            print(f"{lay:3d} {kn} | {t['fused']:7.3f} ({n / t['fused'] / 1e3:7.1f}) | {nP / t['primary'] / 1e3:7.1f} | "
                  f"{nB / t['bounce'] / 1e3:7.1f} | {nS / t['shadow'] / 1e3:7.1f}", flush=True)


if __name__ == "__main__":
    main()
