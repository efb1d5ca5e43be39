#!/usr/bin/env python3
"""The one-launch step (nnbvh_trace_batches_device) with its four batches in different orders: the waves drain
batch 0 first, so the launch's tail is the drain of the LAST batch's longest rays.  Interleaved, one process."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nn_bvh_amd import BVHAggregate, HIT_DTYPE, build_tree, make_prims, scene
    verts, tris, source = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    _, px, py = scene.camera_rays("crown", seed=1, sample=0, return_pixels=True)
    tiles = np.lexsort((px, py, px // 4, py // 4))
    primary = np.stack([scene.camera_rays("crown", seed=1, sample=s) for s in range(8)], 1)[tiles].reshape(-1)
    hits = agg.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    bhits = agg.Intersect(bounce)
    bounce2 = scene.bounce_rays(bounce, bhits, verts, tris, seed=4)
    shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    mk = lambda n: torch.empty(n, dtype=torch.uint8, device="cuda")  # noqa: E731
    B = {"P": ("closest", dev(primary), len(primary), mk(len(primary) * 32)),
         "B1": ("closest", dev(bounce), len(bounce), mk(len(bounce) * 32)),
         "B2": ("closest", dev(bounce2), len(bounce2), mk(len(bounce2) * 32)),
         "S": ("any", dev(shadow), len(shadow), mk(len(shadow)))}
    n = sum(b[2] for b in B.values())
    stream = torch.cuda.current_stream().cuda_stream
    orders = [("P", "B1", "B2", "S"), ("B2", "B1", "P", "S"), ("B2", "B1", "S", "P"), ("S", "P", "B1", "B2"),
              ("B1", "B2", "S", "P"), ("P", "S", "B1", "B2"), ("B2", "P", "B1", "S"), ("S", "B2", "B1", "P")]
    times = {o: [] for o in orders}
    for rnd in range(6):
        for o in orders:
            batches = [(B[k][0], B[k][1].data_ptr(), B[k][2], B[k][3].data_ptr()) for k in o]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            agg.trace_batches_device(batches, stream)
            b.record()
            torch.cuda.synchronize()
            if rnd:
                times[o].append(a.elapsed_time(b))
    print(f"# {source}; {n} rays per step")
    for o in orders:
        t = float(np.median(times[o]))
        print(f"{' '.join(o):14s} {t:7.3f} ms  {n / t / 1e3:7.1f} Mray/s", flush=True)
    if "--knobs" in sys.argv:  # scheduling knobs on the bench's order (S B2 B1 P), interleaved: --knobs a=1,2:b=3,4
        import itertools
        spec = sys.argv[sys.argv.index("--knobs") + 1] if len(sys.argv) > sys.argv.index("--knobs") + 1 else \
            "int_repeat=2,3,4:prim_weight=24,32,48:refill_weight=6,8,12"
        keys = [kv.split("=")[0] for kv in spec.split(":")]
        vals = [[int(x) for x in kv.split("=")[1].split(",")] for kv in spec.split(":")]
        o = ("S", "B2", "B1", "P")
        batches = [(B[k][0], B[k][1].data_ptr(), B[k][2], B[k][3].data_ptr()) for k in o]
        combos = list(itertools.product(*vals))
        kt = {c: [] for c in combos}
        for rnd in range(6):
            for c in combos:
                for key, v in zip(keys, c):
                    agg.set_option(key, v)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                agg.trace_batches_device(batches, stream)
                b.record()
                torch.cuda.synchronize()
                if rnd:
                    kt[c].append(a.elapsed_time(b))
        print(" ".join(keys), "| ms")
        for c in sorted(combos, key=lambda c: np.median(kt[c])):
            print(" ".join(f"{v:6d}" for v in c), f"| {np.median(kt[c]):7.3f}", flush=True)


if __name__ == "__main__":
    main()
