#!/usr/bin/env python3
"""Interleaved A/B timing of the scheduling knobs (speed only; results never change) on the
bench workload.  Prints a table of Mray/s per ray class for every combination, median over
rounds, all variants run round-robin in ONE process (cdna_hip_programming.md §5.4 rule 24)."""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--windows", default="8")
    ap.add_argument("--blocks", default="0")
    ap.add_argument("--refill", default="8", help="refill_weight values")
    ap.add_argument("--primat", default="8", help="prim_weight values")
    ap.add_argument("--stats", action="store_true", help="print scheduling stats (NNBVH_STATS build)")
    ap.add_argument("--xcd", default="1")
    ap.add_argument("--minwaves", default="0")
    ap.add_argument("--intrepeat", default="1")
    ap.add_argument("--primrepeat", default="2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=1, help="concatenate each batch this many times")
    args = ap.parse_args()
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    from nn_bvh_amd._lib import HIT_DTYPE

    verts, tris, source = scene.load_scene(args.scene)
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    cam = args.scene if args.scene in scene.CAMERAS else "crown"
    primary = scene.camera_rays(cam, seed=1, sample=0)
    hits = agg.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    lo, hi = verts.min(0), verts.max(0)
    shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
    stream = torch.cuda.current_stream().cuda_stream

    def dev(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()

    if args.repeat > 1:
        primary, bounce, shadow = (np.concatenate([b] * args.repeat) for b in (primary, bounce, shadow))
    d = {"primary": dev(primary), "bounce": dev(bounce), "shadow": dev(shadow)}
    n = {"primary": len(primary), "bounce": len(bounce), "shadow": len(shadow)}
    out = torch.empty(len(primary) * 32, dtype=torch.uint8, device="cuda")
    combos = list(itertools.product([int(x) for x in args.windows.split(",")],
                                    [int(x) for x in args.blocks.split(",")],
                                    [int(x) for x in args.refill.split(",")],
                                    [int(x) for x in args.xcd.split(",")],
                                    [int(x) for x in args.primat.split(",")],
                                    [int(x) for x in args.minwaves.split(",")],
                                    [int(x) for x in args.primrepeat.split(",")],
                                    [int(x) for x in args.intrepeat.split(",")]))
    times = {c: {k: [] for k in d} for c in combos}

    def run(kind):
        if kind == "shadow":
            agg.intersect_p_device(d[kind].data_ptr(), out.data_ptr(), n[kind], stream=stream)
        else:
            agg.intersect_device(d[kind].data_ptr(), out.data_ptr(), n[kind], stream)

    for rnd in range(args.rounds + 1):
        for c in combos:
            agg.set_option("stack_window", c[0])
            agg.set_option("blocks_per_cu", c[1])
            agg.set_option("refill_weight", c[2])
            agg.set_option("xcd_queues", c[3])
            agg.set_option("prim_weight", c[4])
            agg.set_option("int_repeat", c[7])
            agg.set_option("prim_repeat", c[6])
            for kind in d:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run(kind)
                b.record()
                torch.cuda.synchronize()
                if rnd:  # round 0 = warm-up
                    times[c][kind].append(a.elapsed_time(b))
                if args.stats and rnd == args.rounds:
                    st = agg.sched_stats()
                    util = {k: round(st[k + "_lanes"] / max(1, st[k + "_trips"]) / 64, 3)
                            for k in ("int", "prim", "refill")}
                    print(f"  stats {c} {kind}: trips/ray I {st['int_trips'] * 64 / n[kind]:.1f} "
                          f"P {st['prim_trips'] * 64 / n[kind]:.1f} R {st['refill_trips'] * 64 / n[kind]:.2f}"
                          f" lanes/trip {util} during I trips: waiting-on-prim "
                          f"{st['i_nprim'] / max(1, st['int_trips']):.1f} idle "
                          f"{st['i_nidle'] / max(1, st['int_trips']):.1f}; cycles/trip I "
                          f"{st['int_cycles'] / max(1, st['int_trips']):.0f} P "
                          f"{st['prim_cycles'] / max(1, st['prim_trips']):.0f} R "
                          f"{st['refill_cycles'] / max(1, st['refill_trips']):.0f}; share of wave time I "
                          f"{st['int_cycles'] / max(1, st['int_cycles'] + st['prim_cycles'] + st['refill_cycles']):.2f} P "
                          f"{st['prim_cycles'] / max(1, st['int_cycles'] + st['prim_cycles'] + st['refill_cycles']):.2f}; "
                          f"interior steps/trip {st['int_steps'] / max(1, st['int_trips']):.2f} lanes/step "
                          f"{st['int_step_lanes'] / max(1, st['int_steps']):.1f}")
                elif args.stats:
                    agg.sched_stats()
    print(f"# {source}; rays: {n}")
    print("window blocks refillw xcd primw minw prep irep | primary bounce shadow  Mray/s (median)")
    for c in combos:
        r = [n[k] / np.median(times[c][k]) / 1e3 for k in ("primary", "bounce", "shadow")]
        print(f"{c[0]:6d} {c[1]:6d} {c[2]:6d} {c[3]:3d} {c[4]:6d} {c[5]:4d} {c[6]:2d} {c[7]:3d} | {r[0]:7.1f} {r[1]:7.1f} {r[2]:7.1f}", flush=True)


if __name__ == "__main__":
    main()
